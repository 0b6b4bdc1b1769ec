"""CPU oracle for the Open-o3-Video generate path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``open_o3_video_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

Every function restates (in plain numpy / torch-CPU, never importing
``transformers`` or the reference) the algorithm the reference executes for the
hot path, citing the file:line it follows:

* ``R:``  = /root/reference/...            (marinero4972/Open-o3-Video)
* ``TF:`` = transformers 5.15.0 (third-party; the reference pins commit
  336dc69d of the same library, R:setup.sh:4) -- this is where the model
  arithmetic actually lives.

Pinning: the oracle is checked in ``tests/test_oracle_golden.py`` against
golden vectors generated in the build container by ``tools/make_golden.py``
from (a) the reference's own importable pure functions and (b) the installed
``transformers`` Qwen2.5-VL model code with seeded weights.  The reference
ships no tests or golden data of its own.
"""
