"""CPU oracle for the SRCGAN training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``srcgan_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
modules from ``/root/reference/src`` on CPU in the build container and stores
their inputs / outputs / gradients as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors
(``make_golden_legacy.py`` for the legacy generators of model/model.py:347-440).
The full-cycle generator G_B (``RDDBNetA``) has no source in the reference
(SURVEY.md section 8a-10), so that one component is "parity unpinned" and is
checked only against this restatement.
"""
from .srcgan_oracle import *  # noqa: F401,F403
