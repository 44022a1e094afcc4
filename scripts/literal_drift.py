"""Writes profiles/<tag>_literal_drift.json: the drift of the oracle's numerical specification against the #[cube] sources
taken literally (oracle/drift.py; bounds asserted by tests/test_oracle_literal_drift.py).  CPU only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from brush_amd import synth  # noqa: E402
from oracle import drift  # noqa: E402
import util  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
report = {"what": "specification build of the oracle vs BO_LITERAL builds (1: libm exp/log/atan2, no fma but calc_sigma's; 2: no fma at all)", "cases": {}}
for case in ("tiny_case", "basic_case"):
    sc, ref = util.golden_case(case)
    h, w = ref.shape[:2]
    for variant in ("literal1", "literal2"):
        report["cases"]["%s/%s" % (case, variant)] = drift.measure(sc, util.golden_camera_params(w, h), variant, bg=(0.0, 0.0, 0.0))
for sh in (0, 3):
    sc, w, h = synth.config_scene("1m_1080p", sh)
    cp = synth.default_camera_params(w, h)
    v = (np.random.default_rng(17 + sh).uniform(-1, 1, (h, w, 4)) / (h * w)).astype(np.float32)
    spec = drift.render(sc, cp, "spec", v)
    for variant in ("literal1", "literal2"):
        report["cases"]["1m_1080p_sh%d/%s" % (sh, variant)] = drift.measure(sc, cp, variant, v_output=v, spec=spec)
out = os.path.join(ROOT, "profiles", "%s_literal_drift.json" % tag)
json.dump(report, open(out, "w"), indent=1)
print(out)
