#!/bin/bash
# Static figures of the walk kernels (no GPU needed): registers, LDS, spills, instruction lines per kernel.
#   usage: bash tools/isa_stats.sh [extra hipcc flags, e.g. -DSPX_PTR_NT]      (assembly left in /tmp/isa/walk.s)
mkdir -p /tmp/isa
[ -n "$ISA_REUSE" ] || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -S --cuda-device-only "$@" \
  -o /tmp/isa/walk.s "$(dirname "$0")/../spumoni_amd/csrc/spx_walk.hip" 2>/dev/null
python3 - <<'PY'
import re
s = open('/tmp/isa/walk.s').read()
names = {'0': 'PML', '1': 'MS'}
for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+)\n(?:(?!\.name:).)*?\.name:\s+(_ZN3spx\S*k_walk_fastILi(\d)ELb(\d)ELb(\d)ELi(\d)E\S*)\n(.*?)\.wavefront_size', s, re.S):
    lds, sym, mode, doc, narrow, chunk, body = m.groups()
    g = lambda k: lds if k == 'group_segment_fixed_size' else re.search(r'\.%s:\s+(\d+)' % k, body).group(1)
    # instruction lines of the kernel's text
    i0 = s.index('\n' + sym + ':')
    i1 = s.index('.Lfunc_end', i0)
    ins = [l for l in s[i0:i1].split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    kinds = lambda p: sum(1 for l in ins if l.strip().startswith(p))
    print(f"k_walk_fast<{names[mode]:3s} doc={doc} narrow={narrow} chunk={chunk}>  vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} "
          f"spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')}  lds {g('group_segment_fixed_size'):>6s} B  instr {len(ins):5d} "
          f"(v_ {kinds('v_')}, s_ {kinds('s_')}, ds_ {kinds('ds_')}, global_ {kinds('global_')})")
PY
