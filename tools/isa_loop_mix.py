#!/usr/bin/env python3
"""Instruction mix of the hot loop of a kernel in a gfx950 assembly listing (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -c slowfast_amd/csrc/sf_api.hip --cuda-device-only -S -o /tmp/sf.s
    python tools/isa_loop_mix.py /tmp/sf.s <mangled-kernel-name-prefix> ...

For every kernel it locates the largest cluster of v_pk_fma_f32 (the unrolled FMA body), extends it back to the label
that contains the global loads feeding it and forward to the loop's branch, and counts instructions by class; NumVgprs /
Occupancy are read from the kernel's metadata comments.  Used to compare kernel generations when no MI355X is at hand."""
import re,sys
from collections import Counter
src=open(sys.argv[1]).read().split("\n")
for name in sys.argv[2:]:
    i0=next(i for i,l in enumerate(src) if l.startswith(name) and ":" in l)
    i1=next(i for i in range(i0,len(src)) if "; Occupancy:" in src[i])
    ls=src[i0:i1]
    pk=[i for i,l in enumerate(ls) if "v_pk_fma_f32" in l]
    # innermost loop: nearest label before first pk that is a loop header, nearest branch after last pk
    first,last=pk[0],pk[-1]
    # restrict to the largest cluster: take pk's within contiguous region (gap<80 lines)
    cl=[pk[0]]
    best=[]
    for a in pk[1:]:
        if a-cl[-1]<80: cl.append(a)
        else:
            if len(cl)>len(best): best=cl
            cl=[a]
    if len(cl)>len(best): best=cl
    first,last=best[0],best[-1]
    start=max(i for i in range(first) if re.match(r"\.LBB\d+_\d+:",ls[i]))
    # walk back to include loads preceding (up to previous label containing global_load)
    while not any("global_load" in l for l in ls[start:first]):
        start=max(i for i in range(start) if re.match(r"\.LBB\d+_\d+:",ls[i]))
    end=next(i for i in range(last,len(ls)) if re.match(r"\s+s_(c)?branch",ls[i]))
    body=ls[start:end+1]
    ops=[m.group(1) for l in body for m in [re.match(r"\s+([a-z]+_[a-z0-9_]+)",l)] if m]
    c=Counter(ops)
    v=sum(n for k,n in c.items() if k.startswith("v_"))
    meta=" ".join(l.strip("; ").strip() for l in src[i0:i1+1] if "NumVgprs:" in l or "; Occupancy:" in l)
    print(name[4:64],meta,"| loop lines",len(body),"VALU",v,"pk_fma",c["v_pk_fma_f32"],"cvt",c["v_cvt_f32_f16_e32"]+c["v_cvt_f32_f16_sdwa"],"cndmask",c["v_cndmask_b32_e32"]+c["v_cndmask_b32_e64"],"mov",c["v_mov_b32_e32"]+c["v_mov_b64_e32"],"SALU",sum(n for k,n in c.items() if k.startswith("s_")),"gload",c["global_load_dwordx4"],"ds",sum(n for k,n in c.items() if k.startswith("ds_")))
