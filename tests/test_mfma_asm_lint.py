"""CPU: the compiled kernels are free of the hazards hipcc cannot pad for inline-asm MFMAs (tools/check_mfma_hazard.py): a vector
instruction writing an A / B / C operand register right in front of an `asm volatile("v_mfma...")`, something reading such an MFMA's
result before it has settled, an MFMA multiplying by it too soon.  Found in round 4 as run-to-run differences in the last bits of the
DQN gradient (an operand tuple reassembled by v_mov after its registers had been pinned one by one).  The build runs the same lint
(csrc/Makefile); here it is also required to have LOOKED at something: every kernel file with asm MFMAs reports how many it inspected."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

# file -> (a kernel that must be among those inspected, the least number of asm MFMAs the file is known to hold)
EXPECT = {"mlp_mfma.hip": ("mlp_fused_step_kernel", 500), "dqn_mfma.hip": ("dqn_chain_kernel", 1000),
          "mlp_fused_h2.hip": ("mlp_fused_step_h2_kernel", 250)}


@pytest.mark.parametrize("src", sorted(EXPECT))
def test_no_hazard_around_an_asm_mfma_and_the_lint_saw_them(src):
    import check_mfma_hazard as lint
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    found, counts = lint.check(os.path.join(REPO, "fly_bproject_amd", "csrc", src))
    assert not found, found[:4]
    kernel, least = EXPECT[src]
    assert any(kernel in k for k in counts), sorted(counts)
    assert sum(counts.values()) >= least, counts
    # the one-launch rollout runs the fused step's chain GEMMs (policy_tile_fs): its kernel must be among the inspected ones too
    if src == "dqn_mfma.hip":      # both arithmetics of the fused DQN update
        assert any("dqn_chain_h2_kernel" in k and n >= 300 for k, n in counts.items()), sorted(counts.items())
    if src == "mlp_mfma.hip":
        assert any("rollout_all_fs_kernel" in k and n > 0 for k, n in counts.items()), sorted(counts)
