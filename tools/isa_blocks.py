#!/usr/bin/env python3
"""Per-basic-block instruction statistics of a kernel's gfx950 assembly (no GPU needed): which blocks are the node routines of
the headline loop, and what they cost.

    hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fno-fast-math --cuda-device-only -S libldpc_amd/csrc/kernels_fused.hip -o /tmp/kf.s
    awk '/^_ZN8ldpc_amd12_GLOBAL__N_118decode_fused_small/,/s_endpgm/' /tmp/kf.s > /tmp/kf_small.s
    python tools/isa_blocks.py /tmp/kf_small.s [first_line last_line min_instructions]

Columns: instructions, VALU, binary64 VALU, reciprocals, LDS reads / writes (a ds_read2 counts two), SALU, scalar and vector
memory instructions, branch targets.  A routine is recognised by its LDS signature: the pair call of degree-4 check nodes with
a leaf reads and writes 6 messages with 4 reciprocals, the degree-3 pair without a leaf 6 with 2, the two-block degree-2
variable-node call 4 and 4 without a reciprocal, the degree-15 block 15 and 15 with one (profiles/r4_isa_budget.md)."""
import re,sys
lines=open(sys.argv[1]).read().splitlines()
lo=int(sys.argv[2]) if len(sys.argv)>2 else 0
hi=int(sys.argv[3]) if len(sys.argv)>3 else len(lines)
blocks=[];cur=None
for i,l in enumerate(lines):
    if i<lo or i>=hi: continue
    t=l.strip()
    m=re.match(r'^(\.LBB\d+_\d+):',t)
    if m or cur is None:
        cur={'name':m.group(1) if m else 'entry','start':i+1,'n':0,'valu':0,'rcp':0,'dsr':0,'dsw':0,'salu':0,'smem':0,'vmem':0,'f64':0,'br':[]}
        blocks.append(cur)
        if m: continue
    if not t or t.startswith((';','.','//')): continue
    op=t.split()[0]
    cur['n']+=1
    if op.startswith('v_'):
        cur['valu']+=1
        if 'f64' in op: cur['f64']+=1
        if op.startswith('v_rcp_f64'): cur['rcp']+=1
    elif op.startswith('ds_read'): cur['dsr']+= 2 if 'read2' in op else 1
    elif op.startswith('ds_write'): cur['dsw']+= 2 if 'write2' in op else 1
    elif op.startswith('s_load'): cur['smem']+=1
    elif op.startswith(('global_','buffer_','scratch_','flat_')): cur['vmem']+=1
    elif op.startswith('s_'):
        cur['salu']+=1
        if op.startswith(('s_cbranch','s_branch')): cur['br'].append(t.split()[-1])
for b in blocks:
    if b['n']>=int(sys.argv[4]) if len(sys.argv)>4 else 1:
        print(f"{b['name']:12s} L{b['start']:5d} n={b['n']:4d} valu={b['valu']:4d} f64={b['f64']:3d} rcp={b['rcp']:2d} dsr={b['dsr']:2d} dsw={b['dsw']:2d} salu={b['salu']:3d} smem={b['smem']} vmem={b['vmem']} -> {','.join(b['br'])}")
