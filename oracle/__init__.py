"""CPU oracle for the hydra-pspec Gibbs hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``hydra_pspec_amd/`` may import this
package: it is the checker for the HIP path (tests/, ``__graft_entry__.smoke``)
and the timed ``cpu_baseline`` leg of ``bench.py``; it is never the thing that
is shipped or measured as the product.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
real reference (``/root/reference/hydra_pspec``) in the build container with
``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` checks every
function here against those vectors.
"""
from . import pspec_ref, dpss_ref, oqe_ref  # noqa: F401
