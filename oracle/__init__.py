"""CPU oracle for the stitching hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 for the floating-point
network, numpy / plain C for the integer + index arithmetic) of the
reference's ``FlowHomoAdpater.forward`` path.  Every function cites the
reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything from here, and only as the checker.
The product package never imports it: the product fails loudly when the HIP
extension is missing instead of falling back to this code.

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned against outputs of the
reference's own Python, imported on CPU in the build container with the
third-party stubs of ``oracle/ref_harness`` and driven with the seeded
weights of ``oracle.spec.seeded_state_dict``; the resulting vectors are
committed under ``tests/golden`` together with the script that made them
(``oracle/ref_harness/make_goldens.py``).  Third-party arithmetic that is not
in the reference tree (timm 0.4.12 Twins-SVT, torchvision 0.13 ResNet-50 /
Resize) is restated from its published definition: parity against those
libraries themselves is unpinned (they are not installable here).
"""
