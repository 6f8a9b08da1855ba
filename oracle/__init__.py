"""CPU oracle for the DSRL stage-3 training hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It is a numpy restatement of what the reference computes on
the path SURVEY.md §8 scopes (models/DSRL.py head, models/modules/ASPP.py, models/losses/FALoss.py and
the loss mix / SGD step of command_handlers/train_or_resume.py:435-445).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and only as the
checker (or the timed CPU baseline) - never as a product code path.  The product
(dualsuperreslearningforsemseg_amd) never imports `oracle` and fails loudly without its HIP library.

Parity pinning: the reference ships no tests or golden vectors, so the oracle is pinned
against outputs of the reference's own modules imported in the build container; the generating script
is tests/golden/make_golden.py and the vectors live in tests/golden/*.npz
(tests/test_oracle_vs_golden.py checks every one of them).
"""
from .dsrl_oracle import *      # noqa: F401,F403
from . import philox            # noqa: F401
