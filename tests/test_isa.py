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
CSRC = os.path.join(REPO, 'stc-gnn_amd', 'csrc')


def _library_flags(source):
    """The flags libstc_hip.so compiles ``source`` with (csrc/Makefile: common flags + the file's EXTRA_*), as a list."""
    out = subprocess.check_output(['make', '-s', '-C', CSRC, 'flags-' + os.path.splitext(source)[0]], text=True)
    flags = out.split()
    assert '--offload-arch=gfx950' in flags and '-O3' in flags, out
    return flags


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not present')
@pytest.mark.parametrize('source,n_kernels', [('stc_cell_bwd_x3.hip', 5), ('stc_cell_bwd_x3_acc2.hip', 1), ('stc_node_x3.hip', 6)])
def test_no_scratch_access_inside_the_node_loops(tmp_path, source, n_kernels):
    out = tmp_path / (source + '.s')
    subprocess.check_call([HIPCC, *_library_flags(source), '-S', '--cuda-device-only', '-o', str(out), os.path.join(CSRC, source)], stderr=subprocess.DEVNULL)
    report = subprocess.check_output(['python3', os.path.join(REPO, 'tools', 'isa_scratch.py'), str(out), 'FmtH2'], text=True)
    kernels = re.split(r'^(_Z\S+)\n', report, flags=re.M)[1:]                 # name, lines, name, lines, ...
    assert len(kernels) // 2 >= n_kernels, report
    bad = []
    for name, lines in zip(kernels[0::2], kernels[1::2]):
        loops = re.findall(r'node loop(?: \d+/\d+)?: (\d+) instr, (\d+) mfma, (\d+) scratch ops', lines)
        assert loops, (name, lines)
        # (the fp16 x 2 backward kernels hold their body twice -- first pass / later passes: every copy's node loop must be clean)
        bad += [(name[:90], int(n_scr)) for n_instr, n_mfma, n_scr in loops if int(n_mfma) > 0 and int(n_scr) > 0]
    assert not bad, bad
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not present')
def test_one_launch_bf16_cell_backward_fits_the_register_file(tmp_path):
    """csrc/stc_node_bf16.hip, cell_bwd_bf16_kernel: one wave per SIMD with the next node's operands requested a node ahead.  The C = 32 forms
    must not touch scratch; the C = 64 wide form is at the edge of the 512-register file (2 scratch accesses per node as built: bounded here,
    so that a change that tips it over -- the first version had 22 -- is seen); the C = 64 narrow form (62 per node) must not be built at all."""
    out = tmp_path / 'bf16.s'
    subprocess.check_call([HIPCC, *_library_flags('stc_node_bf16.hip'), '-S', '--cuda-device-only', '-o', str(out), os.path.join(CSRC, 'stc_node_bf16.hip')],
                          stderr=subprocess.DEVNULL)
    report = subprocess.check_output(['python3', os.path.join(REPO, 'tools', 'isa_scratch.py'), str(out), 'cell_bwd_bf16_kernel'], text=True)
    kernels = re.split(r'^(_Z\S+)\n', report, flags=re.M)[1:]
    got = {}
    for name, lines in zip(kernels[0::2], kernels[1::2]):
        form = re.search(r'cell_bwd_bf16_kernelILi(\d)ELi(\d)E', name).groups()
        got[form] = max(int(n) for n in re.findall(r'node loop(?: \d+/\d+)?: \d+ instr, \d+ mfma, (\d+) scratch ops', lines))
    assert set(got) == {('1', '0'), ('1', '1'), ('2', '0')}, got
    assert got[('1', '0')] == 0 and got[('1', '1')] == 0 and got[('2', '0')] <= 4, got
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not present')
def test_patch_aggregation_keeps_two_workgroups_per_cu_and_an_lds_only_barrier(tmp_path):
    """csrc/stc_spmm_patch.hip is built so that a chunk's result stores stay in flight under the next chunk's staging and two workgroups share a
    compute unit.  Checked in the assembly of all 24 forms (6 table widths x {plain, Y0} x {fp32, bf16}): at most 256 registers (a refactoring
    that cost 16 registers once halved the occupancy: 200 -> 270 us), no scratch for tables of up to 16 entries per row, and every s_barrier
    preceded by ``s_waitcnt lgkmcnt(0)`` alone -- ``__syncthreads()`` would put ``vmcnt(0)`` there, a wait for the previous chunk's stores
    (HISTORY section 10)."""
    out = tmp_path / 'patch.s'
    subprocess.check_call([HIPCC, *_library_flags('stc_spmm_patch.hip'), '-S', '--cuda-device-only', '-o', str(out), os.path.join(CSRC, 'stc_spmm_patch.hip')],
                          stderr=subprocess.DEVNULL)
    text = out.read_text()
    bodies = re.findall(r'^(_ZN\S*spmm_patch_kernelILi(\d+)ELb(\d)ELb(\d)E\S*):[^\n]*\n(.*?)s_endpgm', text, flags=re.M | re.S)
    assert len(bodies) == 24, len(bodies)
    for name, width, has_y0, bf16, body in bodies:
        if int(width) <= 16:                                           # (tables of 24 / 32 entries per row spill a few registers: rare shapes)
            assert 'scratch_' not in body, name
        regs = re.search(r'\.amdhsa_kernel ' + re.escape(name) + r'\n.*?\.amdhsa_next_free_vgpr (\d+)', text, flags=re.S)
        assert regs and int(regs.group(1)) <= 256, (name, 'more than 256 registers: one workgroup per compute unit')
        lines = [ln.strip() for ln in body.split('\n') if ln.strip() and not ln.strip().startswith(';')]
        barriers = [i for i, ln in enumerate(lines) if ln == 's_barrier']
        assert len(barriers) >= 2, name
        assert all(lines[i - 1] == 's_waitcnt lgkmcnt(0)' for i in barriers), (name, [lines[i - 1] for i in barriers])
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not present')
def test_two_ring_kernels_keep_two_workgroups_per_cu_without_scratch(tmp_path):
    """csrc/stc_spmm_ring2.hip: two workgroups per compute unit (48 KiB tile each) need <= 256 registers; the forms the metric step and
    configuration 4 dispatch -- the sum with up to two addends and one gathered plane, the blend, every chain form -- must not touch scratch
    (a version with a deeper operand ring spilled 16 accesses per chunk and ran at 1 004 instead of 590 us); barriers wait for LDS alone."""
    out = tmp_path / 'ring2.s'
    subprocess.check_call([HIPCC, *_library_flags('stc_spmm_ring2.hip'), '-S', '--cuda-device-only', '-o', str(out), os.path.join(CSRC, 'stc_spmm_ring2.hip')],
                          stderr=subprocess.DEVNULL)
    text = out.read_text()
    bodies = re.findall(r'^(_ZN\S*ring2_sum_kernelILi(\d)ELb(\d)ELi(\d)ELi(\d)E\S*):[^\n]*\n(.*?)s_endpgm', text, flags=re.M | re.S)
    assert len(bodies) == 12 + 1 + 30, len(bodies)                     # sum: {A, A + A2} x 0..5 addends; blend; chain: {A, A + A2} x 0..2 x 1..5
    for name, mode, dual, n_add, n0, body in bodies:
        regs = re.search(r'\.amdhsa_kernel ' + re.escape(name) + r'\n.*?\.amdhsa_next_free_vgpr (\d+)', text, flags=re.S)
        assert regs and int(regs.group(1)) <= 256, (name, 'more than 256 registers: one workgroup per compute unit')
        if mode != '0' or (dual == '0' and int(n_add) <= 2):
            assert 'scratch_' not in body, name
        lines = [ln.strip() for ln in body.split('\n') if ln.strip() and not ln.strip().startswith(';')]
        barriers = [i for i, ln in enumerate(lines) if ln == 's_barrier']
        assert len(barriers) >= 4, name
        assert all(lines[i - 1] == 's_waitcnt lgkmcnt(0)' for i in barriers), (name, [lines[i - 1] for i in barriers])
    shutil.rmtree(tmp_path, ignore_errors=True)
