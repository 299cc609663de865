#!/usr/bin/env python3
"""Dev tool: where does the time of the twelve-table GPU parity test go (oracle vs GPU vs fixtures)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkm_amd
from oracle.oracle_py import Oracle
from tests import cpu_fixtures as CF

def T(label, f):
    t = time.time(); r = f(); print("%-40s %.2f s" % (label, time.time() - t), flush=True); return r

o = Oracle()
tables, ctls = T("build_full_segment", lambda: CF.build_full_segment(o))
ctx = T("Context", lambda: zkm_amd.Context(0))
want = T("oracle.prove_with_traces", lambda: o.prove_with_traces(tables, ctls, public_values=[1, 2, 3]))
got = T("ctx.prove_with_traces (cold)", lambda: ctx.prove_with_traces(tables, ctls, public_values=[1, 2, 3]))
got = T("ctx.prove_with_traces (warm)", lambda: ctx.prove_with_traces(tables, ctls, public_values=[1, 2, 3]))
assert (got[0] == want[0]).all()
T("oracle.verify_all", lambda: o.verify_all(tables, ctls, got[0], got[1], public_values=[1, 2, 3]))
img = T("segment_image", lambda: zkm_amd.segment_image(tables, ctls, public_values=[1, 2, 3]))
T("ctx.prove_segment_image", lambda: ctx.prove_segment_image(img))
