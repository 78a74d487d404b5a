"""Instruction mix of one kernel in a `hipcc -save-temps` .s file: register/scratch figures and per-opcode counts, for the whole kernel
and for its largest loop body (the text between a label and the backward branch to it).  Usage: isa_mix.py file.s kernel_substring"""
import re, sys
from collections import Counter

def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    m = re.search(r'^(\S*' + re.escape(key) + r'\S*):\s*;', s, re.M)
    if not m:
        sys.exit("kernel not found")
    end = s.index('.end_amdhsa_kernel', m.end())
    body = s[m.end():end]
    for k in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_private_segment_fixed_size'):
        v = re.search(re.escape(k) + r'\s+(\d+)', body)
        print(k, v.group(1) if v else None)
    text = body[:body.index('.section') if '.section' in body else len(body)]
    lines = text.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    best = (0, 0, 0)
    for i, l in enumerate(lines):
        mm = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            if i - labels[mm.group(1)] > best[0]:
                best = (i - labels[mm.group(1)], labels[mm.group(1)], i)
    def mix(ls, title):
        ins = [l.split()[0] for l in ls if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        c = Counter(ins)
        print(f"== {title}: {len(ins)} instructions")
        groups = Counter()
        for k, v in c.items():
            g = ('mfma' if 'mfma' in k else 'accvgpr' if 'accvgpr' in k else 'exp' if k.startswith('v_exp') else 'ds' if k.startswith('ds_') else
                 'scratch' if k.startswith('scratch') else 'vmem' if k.startswith(('global_', 'buffer_')) else 'valu' if k.startswith('v_') else
                 'salu' if k.startswith('s_') else 'other')
            groups[g] += v
        print(dict(groups))
        print(', '.join(f"{k} {v}" for k, v in c.most_common(40)))
    mix(lines, 'kernel')
    if best[0]:
        mix(lines[best[1]:best[2] + 1], f'largest loop (lines {best[1]}..{best[2]})')

main()
