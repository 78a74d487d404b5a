"""Debug aid: is the speaker / text KV of item 0 bit-identical when the same item is encoded in a batch of 1 and in a batch of 2?
Usage: python tools/debug_voice.py  (prints max |diff| per cache; set ECHO_GEMM_FORCE=0,1 to rule split-K in or out)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echo_tts_amd as E
from oracle import echo_ref as R
from tests.golden_defs import TINY, tiny_inputs

for dt in (torch.float32, torch.bfloat16):
    w = {k: v.to(dt) for k, v in R.make_dit_weights(TINY, seed=0).items()}
    m = E.EchoDiT(TINY, w, dtype=dt, device="cuda:0")
    ids, tmask, spk, smask, x0 = tiny_inputs(TINY, batch=2)
    spk1, sm1 = spk[:1], torch.ones_like(smask[:1])
    res = {}
    for B in (1, 2):
        m.get_kv_cache_text(ids[:1].expand(B, -1).contiguous(), tmask[:1].expand(B, -1).contiguous())
        kvs = m.get_kv_cache_speaker(spk1.expand(B, -1, -1).contiguous().to(dt), sm1.expand(B, -1).contiguous())
        for l in range(TINY.num_layers):
            k, v = kvs.layer(l)
            res[(B, l)] = (k[0].clone(), v[0].clone())
    for l in range(TINY.num_layers):
        dk = float((res[(1, l)][0] - res[(2, l)][0]).abs().max())
        dv = float((res[(1, l)][1] - res[(2, l)][1]).abs().max())
        print(dt, "layer", l, "speaker K max|diff|", dk, "V", dv)
