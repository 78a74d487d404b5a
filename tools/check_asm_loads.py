"""Audit of a hipcc -save-temps .s: VALU writes into the destination registers of an asm ds_read_b128 that is still in flight (hipcc treats an asm
load as complete at the statement).  Straight-line scan from each read to the counted lgkmcnt wait that retires it.  Usage: check_asm_loads.py [kernel_substring]
(expects /tmp/att/attention-hip-amdgcn-amd-amdhsa-gfx950.s, as written by hipcc -save-temps from /tmp/att)."""
import re,sys
s=open('/tmp/att/attention-hip-amdgcn-amd-amdhsa-gfx950.s').read()
key=sys.argv[1] if len(sys.argv)>1 else 'attn4_kernelILi0E'
m=re.search(r'^(\S*'+key+r'\S*):\s*;', s, re.M)
end=s.index('.end_amdhsa_kernel', m.end())
lines=[l for l in s[m.end():end].split('\n') if l.strip() and not l.strip().startswith(';')]
def regs(tok):
    mm=re.match(r'v\[(\d+):(\d+)\]',tok)
    if mm: return set(range(int(mm.group(1)),int(mm.group(2))+1))
    mm=re.match(r'v(\d+)$',tok)
    return {int(mm.group(1))} if mm else set()
bad=0
for i,l in enumerate(lines):
    if 'ds_read_b128' in l:
        dst=regs(l.split()[1].rstrip(','))
        cnt=0
        for j in range(i+1,min(i+400,len(lines))):
            t=lines[j].split()
            if re.match(r'^\.LBB',lines[j]): break          # straight-line only
            if t[0].startswith(('s_branch','s_cbranch','s_endpgm')): break
            if 'ds_read_b128' in lines[j]:
                cnt+=1
            if 's_waitcnt' in lines[j]:
                mm=re.search(r'lgkmcnt\((\d+)\)',lines[j])
                if mm and int(mm.group(1))<=cnt: break       # in-order return: this read has landed
            if t[0].startswith('v_') and not t[0].startswith(('v_mfma','v_cmp')):
                d=regs(t[1].rstrip(','))
                if d & dst:
                    bad+=1
                    if bad<8: print('CLOBBER', i, l.strip(), '|', j, lines[j].strip())
print('in-flight ds_read destinations written by VALU:',bad)
