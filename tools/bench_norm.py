"""AdaLN-apply norm at the sampler's shapes (bf16, D = 2048): microseconds and GB/s per launch, buffers rotated through 1 GiB so that
every launch reads from HBM.  ECHO_NORM_ROWS=1|2|4 selects rows per wave (1 = the generic kernel)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echo_tts_amd import _lib as L
lib = L.load_library()
dev = "cuda:0"
D = 2048
for M in (15360, 46080, 1920):
    nbuf = max(2, (1 << 30) // (M * D * 2))
    xs = [torch.randn((M, D), device=dev).bfloat16() for _ in range(nbuf)]
    y = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    w0, w1 = torch.randn((D,), device=dev).bfloat16(), torch.randn((D,), device=dev).bfloat16()
    st = torch.cuda.current_stream().cuda_stream
    run = lambda x: L.check(lib.echo_op_norm(L.ECHO_BF16, 0, x.data_ptr(), D, y.data_ptr(), D, M, D, 1e-5, w0.data_ptr(), w1.data_ptr(), st))
    for x in xs[:2]:
        run(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for i in range(n):
        run(xs[i % nbuf])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"ECHO_NORM_ROWS={os.environ.get('ECHO_NORM_ROWS', 'default')} M={M}: {us:.1f} us per launch, {2 * M * D * 2 / us / 1e3:.0f} GB/s (read + write)")
    ref = xs[0].float()
    ref = (ref * torch.rsqrt(ref.pow(2).mean(-1, keepdim=True) + 1e-5) * w0.float() + w1.float()).bfloat16()
    run(xs[0]); torch.cuda.synchronize()
    print("   max |diff| vs torch:", float((y.float() - ref.float()).abs().max()), " mismatching elements:", int((y != ref).sum()))
