"""Static check of the gfx950 ISA hipcc emits for the kernels that store through buffer instructions (round 4): no VMEM store
whose scalar offset is a REGISTER may have its data registers rewritten within three instructions.  hipcc inserts the documented
wait state only when the store's soffset is not a register; on gfx950 such a pair delivered the NEW value in some lanes of ~0.3 %
of the rows of a full-size grouping layer (tools/isa_hazards.py).  Needs hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("src,pattern", [("so3_basis.hip", "so3_basis"), ("inter_mfma.hip", "inter_group_wide_kernelILi[124]ELi2E")])
def test_no_store_with_a_register_soffset_has_its_data_rewritten_right_behind_it(src, pattern):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazards.py"),
                          os.path.join(ROOT, "epn_pointcloud_amd", "csrc", src), pattern],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    flagged = [l for l in out.stdout.splitlines() if "[SGPR soffset]" in l]
    assert not flagged, "\n".join(flagged[:10])
    assert "store(s) with a write of their data registers" in out.stdout       # the scan ran and found the kernels
