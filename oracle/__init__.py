"""CPU oracle for the spatiotemporal-DETR hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 CPU restatement of the reference algorithm
(atonderski/future-object-detection, files future_od/models/{st_detr,paper,
transformer,set_criterion}.py and future_od/utils/od_map.py) plus our own authoring
of the third-party symbols the reference imports but does not vendor
(ConditionalDETR.*, torchvision ResNet).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it,
and only as the checker.  The product path (future-object-detection_amd/) never
imports anything from here and fails loudly if its HIP library is missing.

Parity status
-------------
* The reference's OWN files are pinned: tests/golden/make_golden.py imports them in the
  build container (with oracle.thirdparty registered for the two absent packages) and
  the committed fixtures under tests/golden/ are checked by tests/test_oracle_golden.py.
* The six ConditionalDETR symbols and the torchvision ResNet are absent from
  /root/reference (empty submodule, pin unknown; torchvision not installed): for those
  the oracle is "parity unpinned" -- the goldens pin our reading of their published
  semantics (see oracle/thirdparty.py; cross-checked against the independent DETR-loss
  code shipped inside the `transformers` wheel in tests/test_oracle_thirdparty.py).
"""
