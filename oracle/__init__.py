"""CPU oracle package -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.py)."""
from .oracle import OracleConfig, OracleEnv, OraclePmi, actor_actions, build, greedy_actions, lib, philox4x32_10  # noqa: F401
