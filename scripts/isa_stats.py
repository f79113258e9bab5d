#!/usr/bin/env python3
"""developer tool: instruction mix and resources per kernel from `hipcc -S --cuda-device-only` output:  scripts/isa_stats.py file.s [name filter ...]"""
import re, sys, collections
src = open(sys.argv[1]).read()
want = sys.argv[2:] 
# split on function labels
funcs = re.split(r'\n(_Z\w+):[^\n]*\n', src)
out = {}
for i in range(1, len(funcs), 2):
    name, body = funcs[i], funcs[i+1]
    end = body.find('.end_amdhsa_kernel')
    code = body[:body.find('.Lfunc_end')] if '.Lfunc_end' in body else body
    ins = [l.strip().split()[0] for l in code.splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    meta = {}
    for k in ('.amdhsa_next_free_vgpr', '.amdhsa_group_segment_fixed_size', '.amdhsa_kernarg_size'):
        m = re.search(re.escape(k) + r'\s+(\d+)', body)
        if m: meta[k.split('_',1)[1]] = int(m.group(1))
    m = re.search(r'; ScratchSize: (\d+)', body); 
    if m: meta['scratch'] = int(m.group(1))
    m = re.search(r'; Occupancy: (\d+)', body)
    if m: meta['occ'] = int(m.group(1))
    out[name] = (c, meta, len(ins))
import subprocess
for name, (c, meta, n) in out.items():
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r'\(.*', '', dem)
    if want and not any(w in short for w in want): continue
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    pk = sum(v for k, v in c.items() if k.startswith('v_pk_'))
    fma = sum(v for k, v in c.items() if k.startswith(('v_fma_f32','v_fmac_f32','v_fmamk','v_fmaak')))
    mul = sum(v for k, v in c.items() if k.startswith(('v_mul_f32',)))
    add = sum(v for k, v in c.items() if k.startswith(('v_add_f32','v_sub_f32','v_subrev_f32')))
    ds = sum(v for k, v in c.items() if k.startswith('ds_'))
    buf = sum(v for k, v in c.items() if k.startswith('buffer_'))
    mov = sum(v for k, v in c.items() if k.startswith(('v_mov_b32','v_mov_b64','v_accvgpr')))
    print(f"{short[:70]:70s} n={n} valu={valu} pk={pk} fma={fma} mul={mul} add={add} mov={mov} ds={ds} buf={buf} {meta}")
