"""Performance properties that can be read off the compiled code without a GPU (tools/isa_audit.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc (cross-compiles without a GPU)")
def test_batchnorm_column_sums_keep_their_row_loads_in_flight():
    """Round 4's largest single gain (4.46 -> 4.31 ms per step) was a row loop that had become a chain of dependent round trips
    because loads sat behind null checks inside it.  ``bn_bwd_reduce_kernel`` must have NO loop block whose full
    ``s_waitcnt vmcnt(0)`` follows one or two loads, in any of its twelve instantiations (<Z2, DROP> x the three activation
    layouts of round 6: fp32, bf16, bf16 with an fp32 dy)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_audit
    finally:
        sys.path.pop(0)
    rows = isa_audit.audit(os.path.join(ROOT, "myria3d_amd", "csrc", "bn.hip"), [])
    reduce_rows = [r for r in rows if "bn_bwd_reduce_kernel" in r[3]]
    assert len(reduce_rows) == 12, [r[3] for r in rows]  # <Z2, DROP, IO> = twelve instantiations, each with loop blocks
    for flagged, blocks, fewest, name in reduce_rows:
        assert flagged == 0, (name, flagged, blocks, fewest)


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc (cross-compiles without a GPU)")
def test_complete_neighbourhood_lfa_forward_kernels_keep_their_instruction_diet():
    """Round 5: the level-1 LFA forward kernels are VALU-issue bound (SQ counters: 75 % of the SIMDs' issue slots), so their
    instruction count IS their time.  Read off the ISA (tools/isa_count.py: the kernels are straight-line per wave):
    the complete-neighbourhood kernels hold <= 300 static VALU instructions per wavefront at ch = 8 (128 edges: two centres per
    MFMA tile) and <= 260 at ch = 16 (64 edges) — round 4's kernel: 484 / 523 per 64 edges —, their K = 16 instantiations carry
    NO cross-lane swap (in-lane neighbourhoods) and no exec-mask branch for the -1 padding (at most the bounds checks)."""
    import re
    import subprocess
    import tempfile

    src = os.path.join(ROOT, "myria3d_amd", "csrc", "lfa.hip")
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-w", "--cuda-device-only", "-S",
                        src, "-o", tmp.name], check=True, cwd=os.path.dirname(src))
        text = open(tmp.name).read()
    kernels = {}
    kernels16 = {}  # the same kernels with bf16 activation storage (IOH, round 6)
    for m in re.finditer(r"^(_Z19lfa_fwd_full_kernelILi(\d+)ELi(\d+)ELi0ELb([01])EEv7LfaArgs):.*?\.Lfunc_end", text, re.S | re.M):
        body = [ln.strip().split()[0] for ln in m.group(0).split("\n")[1:] if ln.strip() and not ln.strip().startswith((";", "."))
                and not ln.strip().endswith(":")]
        (kernels16 if m.group(4) == "1" else kernels)[(int(m.group(2)), int(m.group(3)))] = body
    assert (8, 16) in kernels and (16, 16) in kernels and (64, 16) in kernels, sorted(kernels)
    valu = lambda ops: sum(1 for o in ops if o.startswith("v_") and not o.startswith(("v_mfma", "v_accvgpr")))
    assert valu(kernels[(8, 16)]) <= 300, valu(kernels[(8, 16)])
    assert valu(kernels[(16, 16)]) <= 260, valu(kernels[(16, 16)])
    assert valu(kernels[(64, 16)]) <= 270, valu(kernels[(64, 16)])
    for key in ((8, 16), (16, 16), (64, 16)):
        ops = kernels[key]
        assert not any(o.startswith("v_permlane") for o in ops), (key, "cross-lane swaps in a K = 16 kernel")
        assert sum(1 for o in ops if o.startswith("s_and_saveexec")) <= 4, (key, "exec-mask branches")
    # K = 32: exactly the three joins of the two half neighbourhoods (maximum, numerator, denominator) per column tile
    assert sum(1 for o in kernels[(16, 32)] if o.startswith("v_permlane16_swap")) == 3
    # bf16 activation storage: the widening of the gathered rows (shift / mask) and the rounding of the stored outputs
    # (v_cvt_pk_bf16_f32) cost a few instructions per thread, nothing more
    for key, extra in (((8, 16), 24), ((16, 16), 24), ((64, 16), 40)):
        assert key in kernels16, sorted(kernels16)
        assert valu(kernels16[key]) <= valu(kernels[key]) + extra, (key, valu(kernels16[key]), valu(kernels[key]))
        assert not any(o.startswith("v_permlane") for o in kernels16[key]), key


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc (cross-compiles without a GPU)")
def test_wave_autonomous_lfa_backward_has_no_barrier_and_no_spill():
    """Round 5: ``lfa_bwd_small_kernel`` (8 / 16 channels, complete neighbourhoods) is built on three properties that only the
    ISA shows: no workgroup barrier anywhere (a wave owns its LDS rows from the gather to the stores), no register spill (a
    scratch reload behind the prefetch would wait for the next trip's loads), and the encoder weights arriving through
    scalar loads (constant address space) instead of 44 / 88 uniform vector loads per trip.  The edge-row form has no atomic."""
    import re
    import subprocess
    import tempfile

    src = os.path.join(ROOT, "myria3d_amd", "csrc", "lfa_bwd.hip")
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-w", "--cuda-device-only", "-S",
                        src, "-o", tmp.name], check=True, cwd=os.path.dirname(src))
        text = open(tmp.name).read()
    seen = 0
    for ch in (8, 16):
        for edge in (0, 1):
          for io in (0, 1):  # (fp32 / bf16 activation storage: the same structure)
            m = re.search(r"^_Z20lfa_bwd_small_kernelILi%dELb%dELb%dEEv10LfaBwdArgsi:.*?\.Lfunc_end" % (ch, edge, io), text, re.S | re.M)
            assert m, (ch, edge, io)
            ops = [ln.strip().split()[0] for ln in m.group(0).split("\n")[1:] if ln.strip() and not ln.strip().startswith((";", "."))
                   and not ln.strip().endswith(":")]
            assert not any(o == "s_barrier" for o in ops), (ch, edge, io, "workgroup barrier")
            assert not any(o.startswith("scratch_") for o in ops), (ch, edge, io, "register spill")
            assert sum(1 for o in ops if o.startswith("s_load_dword")) >= 6, (ch, edge, io, "encoder weights not on the scalar path")
            atom = sum(1 for o in ops if o.startswith("global_atomic"))
            assert atom == (0 if edge else 16), (ch, edge, io, atom)
            assert sum(1 for o in ops if o.startswith("v_mfma")) == 48, (ch, edge, io)
            seen += 1
    assert seen == 8


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc (cross-compiles without a GPU)")
def test_weight_gradient_trips_issue_every_operand_load_before_the_first_wait():
    """Round 6: the batched weight-gradient kernels spent three quarters of their time in dependent round trips to memory — the
    two halves of a concatenated X row were added inside the load helper of each 4-row step and the helper branched on the
    column layout, so the loads of step d + 1 sat behind a wait for step d (``s_waitcnt vmcnt(1)`` after three loads, DEPTH
    times per trip).  ``wgrad_trips_vec`` / ``wgrad_trips_bf16``: in every loop block that holds a trip's operand loads, no
    wait may ask for an operand load of the SAME block (``vmcnt(N)`` with N below the number of vector loads issued so far
    in the block), in the fp32 and bf16-storage instantiations of the 16-tile class and in the bf16 matrix-core variant."""
    import re
    import subprocess
    import tempfile

    src = os.path.join(ROOT, "myria3d_amd", "csrc", "gemm_direct.hip")
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-w", "--cuda-device-only", "-S",
                        src, "-o", tmp.name], check=True, cwd=os.path.dirname(src))
        text = open(tmp.name).read()
    checked = 0
    for bf, h in ((0, 0), (0, 1), (1, 0), (1, 1)):
        m = re.search(r"^_Z19wgrad2_batch_kernelILi4ELi4ELb%dELb%dEEv10WgradBatch:.*?\.Lfunc_end" % (bf, h), text, re.S | re.M)
        assert m, (bf, h)
        trip_blocks = 0
        for block in re.split(r"^\.LBB\d+_\d+:", m.group(0), flags=re.M)[1:]:
            wide = 0  # operand loads (dwordx4 / dwordx2: the row numbers are dword loads) issued so far in this block
            total = sum(1 for ln in block.split("\n") if re.match(r"\s*buffer_load_dwordx[24]", ln))
            if total < 6:
                continue  # not a trip's load phase
            trip_blocks += 1
            for ln in block.split("\n"):
                if re.match(r"\s*buffer_load_dwordx[24]", ln):
                    wide += 1
                w = re.match(r"\s*s_waitcnt .*vmcnt\((\d+)\)", ln)
                if w and wide < total:  # (behind the last load the waits of the MFMA phase begin)
                    assert int(w.group(1)) >= wide, (bf, h, "a wait for an operand load in front of the next load", ln.strip(), wide)
        assert trip_blocks >= 1, (bf, h)
        checked += 1
    assert checked == 4
