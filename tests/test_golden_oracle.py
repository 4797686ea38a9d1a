"""The C restatement and the test-only wave emulation of the kernel source, replayed against the golden
vectors recorded from the real reference engine (tests/golden/make_golden.py)."""
import os
import pytest
from lockstep import EngineAsEnv, golden_files, replay_golden


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_golden(oracle_lib, path):
    ok, msg = replay_golden(path, lambda **cfg: oracle_lib.OraEnv(**cfg))
    assert ok, msg


@pytest.mark.parametrize("path", golden_files(gpu_capable_only=True), ids=lambda p: os.path.basename(p)[:-4])
def test_kernel_source_emulation_matches_golden(emu_lib, path):
    from agarcl_amd import _capi

    def mk(**cfg):
        cfg = dict(cfg); cfg.setdefault("num_agents", 1)
        return EngineAsEnv(_capi.BatchedEngine, lib=emu_lib, **cfg)
    ok, msg = replay_golden(path, mk)
    assert ok, msg
