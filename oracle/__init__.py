"""CPU oracle for the JSS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/jss_oracle.h for the parity status (pinned against
golden traces captured from the live reference).
"""
from .oracle import OracleEnv, build_oracle, rng_u32, rollout_batch, POLICY_IDS  # noqa: F401
