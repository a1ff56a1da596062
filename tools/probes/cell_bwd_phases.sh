#!/bin/bash
# Phase breakdown of the one-launch cell backward: makes a stamped copy of csrc/stc_cell_bwd_x3.hip (s_memtime at the phase boundaries,
# per-wave sums added into a device array), compiles tools/probes/cell_bwd_phases.hip around it and runs it.  MI355X only.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
python3 - "$R" <<'PY'
import sys
R = sys.argv[1]
src = open(R + '/stc-gnn_amd/csrc/stc_cell_bwd_x3.hip').read()
k0 = src.index('__global__ __launch_bounds__(CB_THREADS, 1) void cell_bwd_x3_kernel(CellBwdArgs a) {')
def ins(s, anchor, text, after=True, start=0):
    i = s.index(anchor, start)
    return (s[:i + len(anchor)] + text + s[i + len(anchor):]) if after else (s[:i] + text + s[i:])
s = src
s = ins(s, 'namespace {\n', '\n__device__ unsigned long long g_stamps[8];\n#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_t[i] += t_ - last_t; last_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)\n')
s = ins(s, '    while (node < a.nodes) {\n', '        STAMP(5);\n')
s = ins(s, '        const int lo = opaque(lane);\n', '        STAMP(0);\n', True, k0)
s = ins(s, '        // =========================================================== gate + blend backward (prologue of the gates convolution)\n', '        STAMP(1);\n', False)
s = ins(s, '        // =========================================================== gates convolution (slab form on the planes), as node_bwd_x3_kernel\n', '        STAMP(2);\n', False)
a = '        X3 qd[2];\n        {\n            f32x4 Qd[NRB][2];'
i = s.index(a, s.index('gates convolution (slab form on the planes)'))
s = s[:i] + '        STAMP(3);\n' + s[i:]
s = ins(s, '        cur = nxt;\n        node = next_node;\n', '        STAMP(4);\n', False)
s = ins(s, '    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0), expcnt / lgkmcnt untouched\n', '    unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0}, last_t = __builtin_amdgcn_s_memtime(), n_nodes = 0;\n')
s = ins(s, '        cur = nxt;\n        node = next_node;\n', '        ++n_nodes;\n')
s = ins(s, '    combine_dw<K, LB, 2, CB_WAVES>(', '    if (lane == 0) { for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[i], acc_t[i]); atomicAdd(&g_stamps[6], n_nodes); atomicAdd(&g_stamps[7], 1ull); }\n', False)
open('/tmp/cell_bwd_stamped.hip', 'w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I$R/include -I$R/stc-gnn_amd/csrc -DSTAMPED_SOURCE='"/tmp/cell_bwd_stamped.hip"' $R/tools/probes/cell_bwd_phases.hip -o /tmp/cell_bwd_phases
/tmp/cell_bwd_phases
