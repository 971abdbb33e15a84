"""Circuit front-ends that emit circuit plans for libshielder_hip (build-time tooling)."""
from .builder import PlanBuilder  # noqa: F401
from .withdraw import build_withdraw_plan, build_deposit_plan, build_new_account_plan, new_account_witness_inputs, NewAccountLayout, WithdrawLayout, withdraw_witness_inputs  # noqa: F401
from .phased import build_phased_plan, phased_witness_inputs  # noqa: F401
