"""Instruction-class counts of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): per kernel the
number of MFMA / VALU / packed VALU / LDS / VMEM / accvgpr moves, and the same for the main loop (the basic block with most MFMAs)."""
import re, sys
from collections import Counter

def classes(lines):
    c = Counter()
    for line in lines:
        m = re.match(r'\s+([a-z_0-9]+)', line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_accvgpr'): c['acc_mov'] += 1
        elif op.startswith('v_pk_'): c['valu_pk'] += 1; c['valu'] += 1
        elif op.startswith('v_'): c['valu'] += 1; c['v_mov'] += op.startswith('v_mov')
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')): c['vmem'] += 1; c['scratch'] += op.startswith('scratch_')
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    return dict(c)

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm', s, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    blocks = re.split(r'\n\.LBB\d+_\d+:[^\n]*', body)
    main = max(blocks, key=lambda b: b.count('v_mfma'))
    print(name)
    print('   kernel   ', classes(body.splitlines()))
    print('   main loop', classes(main.splitlines()))
