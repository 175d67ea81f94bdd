"""Multi-step loss trajectory of the whole LoRA + textual-inversion step (main.py:263-382) on CPU: the host plans driven
through the fp32 op emulation (tests/emu_ops.py) against oracle/step_ref.py - same flow and same code as the real-topology
GPU test (tests/test_real_topology_gpu.py), on the toy topologies, with fp32 tolerances."""
import pytest
import torch

from oracle import unet_ref as U
from tests import emu_ops
from tests.test_real_topology_gpu import TOL_FP32, TOL_FP32_FAITHFUL, _bf16_exact, run_step_and_trajectory

pytest.importorskip("transformers")


@pytest.mark.parametrize("version,B,kinds", [("tiny15", 2, ["tiny_l"]), ("tinyxl", 1, ["tiny_l", "tiny_g"])])
def test_trajectory_cpu(version, B, kinds):
    sd = _bf16_exact(U.init_unet_state(U.CONFIGS[version], seed=0))
    traj = run_step_and_trajectory(version, B, 32 if U.CONFIGS[version]["addition"] else 16, sd, kinds, device="cpu", ops=emu_ops,
                                   act_dtype=torch.float32, tol=TOL_FP32, tol_faithful=TOL_FP32_FAITHFUL, rank=4, n_steps=6)
    assert len(traj) == 6


@pytest.mark.parametrize("version,B,kinds", [("tiny15", 2, ["tiny_l"]), ("tinyxl", 1, ["tiny_l", "tiny_g"])])
def test_trajectory_dora_cpu(version, B, kinds):
    """use_dora (optimizer.py:86-95): magnitudes, column factor, scaled backward operands and the magnitude gradient through the same flow."""
    sd = _bf16_exact(U.init_unet_state(U.CONFIGS[version], seed=0))
    run_step_and_trajectory(version, B, 32 if U.CONFIGS[version]["addition"] else 16, sd, kinds, device="cpu", ops=emu_ops,
                            act_dtype=torch.float32, tol=TOL_FP32, tol_faithful=TOL_FP32_FAITHFUL, rank=4, n_steps=6, dora=True)
