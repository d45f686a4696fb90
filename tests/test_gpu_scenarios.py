"""HIP path vs the same KATs, with the oracle cross-checked after every call (DualEnv), plus the real-log replay."""
import pytest

from tests.env_adapters import DualEnv, GpuEnv
from tests.scenarios import SCENARIOS, SCENARIOS_3P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sc", SCENARIOS + SCENARIOS_3P, ids=lambda f: f.__name__)
def test_gpu_scenario(sc):
    sc(lambda **kw: DualEnv(**kw))


def test_gpu_replays_reference_log(golden_dir):
    from tests.test_oracle_replay import replay_all

    replay_all(lambda: GpuEnv(game_mode=2, seed=1), golden_dir)
