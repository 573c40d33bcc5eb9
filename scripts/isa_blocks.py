"""Instruction-class counts per barrier interval of one kernel of a device assembly file (hipcc --cuda-device-only -S):
    python scripts/isa_blocks.py /tmp/conv.s conv_igemm_hp8_kernelILi256ELi7
prints, for every stretch between two s_barrier (and every label), how many MFMA / ds_read / LDS-DMA / VALU / SALU / waitcnt it holds."""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M)
if not m:
    sys.exit("kernel not found")
def cls(l):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'): return None
    if l.endswith(':'): return 'LABEL'
    op = l.split()[0]
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_read'): return 'dsr'
    if op.startswith('ds_write'): return 'dsw'
    if op.startswith('buffer_load') and 'lds' in l: return 'dma'
    if op.startswith('buffer_') or op.startswith('global_'): return 'vmem'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_barrier'): return 'BAR'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_'): return 'salu'
    return op
cur = {}
for l in m.group(2).split('\n'):
    c = cls(l)
    if c is None: continue
    if c in ('BAR', 'LABEL'):
        if cur: print(cur)
        cur = {}
        print('----', l.strip())
    else:
        cur[c] = cur.get(c, 0) + 1
print(cur)
