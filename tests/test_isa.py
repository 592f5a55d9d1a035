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
    ``s_waitcnt vmcnt(0)`` follows one or two loads, in any of its four instantiations."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_audit
    finally:
        sys.path.pop(0)
    rows = isa_audit.audit(os.path.join(ROOT, "myria3d_amd", "csrc", "bn.hip"), [])
    reduce_rows = [r for r in rows if "bn_bwd_reduce_kernel" in r[3]]
    assert len(reduce_rows) == 4, [r[3] for r in rows]  # <Z2, DROP> = four instantiations, each with loop blocks
    for flagged, blocks, fewest, name in reduce_rows:
        assert flagged == 0, (name, flagged, blocks, fewest)
