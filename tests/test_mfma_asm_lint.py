"""CPU: the compiled kernels are free of the one hazard hipcc cannot pad for inline-asm MFMAs (tools/check_mfma_hazard.py): a vector
instruction writing an A / B operand register right in front of an `asm volatile("v_mfma...")`.  Found in round 4 as run-to-run
differences in the last bits of the DQN gradient (an operand tuple reassembled by v_mov after its registers had been pinned one by one)."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


@pytest.mark.parametrize("src", ["mlp_mfma.hip", "dqn_mfma.hip"])
def test_no_vector_write_in_front_of_an_asm_mfma_operand(src):
    import check_mfma_hazard as lint
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    found = lint.check(os.path.join(REPO, "fly_bproject_amd", "csrc", src))
    assert not found, found[:4]
