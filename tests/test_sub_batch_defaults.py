"""vec_env.default_sub_batches: how many free-running arena ranges a batch of a configuration becomes when the caller does not say (host logic only).
The measurements behind the rule: profiles/r06_bench_driver20_full.json (<workload>/pipe4 rows) and profiles/r06_vec_pipe_ab.txt (why the full-batch
step() of AgarioVectorEnv stays ONE range)."""
from agarcl_amd.vec_env import default_sub_batches


def test_default_sub_batches_follow_the_workload():
    assert default_sub_batches(4096) == 1                                     # quiet: one mass-25 agent per arena (C2, tasks 1-4)
    assert default_sub_batches(4096, mode_number=3) == 1
    assert default_sub_batches(4096, mode_number=6) == 4 and default_sub_batches(4096, mode_number=5) == 4      # agents start at mass 1000
    assert default_sub_batches(4096, num_bots=1, mode_number=7) == 4          # tasks 7-10: a bot
    assert default_sub_batches(4096, num_agents=2) == 4
    assert default_sub_batches(512, mode_number=6) == 1                       # a range below 256 arenas cannot fill the chip
    assert default_sub_batches(1024, mode_number=6) == 4
