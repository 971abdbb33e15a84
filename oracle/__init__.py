"""CPU oracle (TEST INFRASTRUCTURE).  See oracle/oracle.h.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (zkos-monorepo_amd/) never does.
"""
from .binding import *  # noqa: F401,F403
