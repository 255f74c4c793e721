"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A plain CPU restatement (torch-CPU fp32 for the floating-point layers, NumPy f64 for
decode / polyline assembly, C for the rasteriser) of the reference's inference hot path
(SURVEY.md §8a rows a1-a9).  Every function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / the timed CPU baseline.  The product
package ``lanemapping_amd`` never imports it; its ops raise if the HIP library is missing.

Pinning status:
  * a3-a7, a9 (FPN, ViT, head, decode, polyline assembly, Segmentor decode): pinned against
    golden vectors produced by importing the reference itself in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz).
  * a2 (LAS->BEV rasteriser): the reference has no rasteriser -> PARITY UNPINNED;
    anchored only by the round-trip through the reference's inverse transform.
"""
