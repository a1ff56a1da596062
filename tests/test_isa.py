"""The split-operand kernels' node loops must not touch scratch.  A spilled register that is reloaded inside the loop is a memory round trip
per node in kernels that run one or two waves per SIMD: eleven reloads of spilled plane addresses made the order-3 gates backward 27 % slower
(round 4) while every basic block WITH matrix instructions was clean -- so the check scans the whole loop (tools/isa_scratch.py).
Compiles the two kernel files to assembly with the library's own flags (~70 s on the CPU box; needs hipcc, no GPU)."""
import os
import re
import shutil
import subprocess

import pytest

from tests.conftest import REPO

HIPCC = '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not present')
@pytest.mark.parametrize('source', ['stc_cell_bwd_x3.hip', 'stc_node_x3.hip'])
def test_no_scratch_access_inside_the_node_loops(tmp_path, source):
    out = tmp_path / (source + '.s')
    csrc = os.path.join(REPO, 'stc-gnn_amd', 'csrc')
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', f'-I{os.path.join(REPO, "include")}', '-fno-slp-vectorize',
                           '-S', '--cuda-device-only', '-o', str(out), os.path.join(csrc, source)], stderr=subprocess.DEVNULL)
    report = subprocess.check_output(['python3', os.path.join(REPO, 'tools', 'isa_scratch.py'), str(out), 'FmtH2'], text=True)
    kernels = re.findall(r'^(_Z\S+)\n\s+node loop: (\d+) instr, (\d+) mfma, (\d+) scratch ops', report, flags=re.M)
    assert len(kernels) >= 6, report
    bad = [(name[:90], int(n_scr)) for name, n_instr, n_mfma, n_scr in kernels if int(n_mfma) > 0 and int(n_scr) > 0]
    assert not bad, bad
    shutil.rmtree(tmp_path, ignore_errors=True)
