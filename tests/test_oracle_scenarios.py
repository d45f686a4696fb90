"""Oracle vs the reference's behavioural KATs (tests/scenarios.py cites each reference test)."""
import pytest

from tests.env_adapters import OracleEnv
from tests.scenarios import SCENARIOS, SCENARIOS_3P, SCENARIOS_ORACLE_ONLY


@pytest.mark.parametrize("sc", SCENARIOS + SCENARIOS_3P + SCENARIOS_ORACLE_ONLY, ids=lambda f: f.__name__)
def test_oracle_scenario(sc):
    sc(lambda **kw: OracleEnv(**kw))
