#!/usr/bin/env python3
"""Diagnostics: host time of one km_batch_run (its ~14 HIP calls) — is the pipelined step bound by
the launching thread?  usage: launch_cost.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from km_amd import lib as kmlib, synth  # noqa: E402

T, L, K = 10000, 500, 31
case = synth.make_case(n_targets=T, length=L, k=K, n_keys=20_000_000, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
offs = np.arange(T + 1, dtype=np.uint64) * np.uint64(L)
bs, sts = [], []
for q in range(4):
    b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
    b.set_targets_packed(blob, offs)
    bs.append(b)
    sts.append(kmlib.stream_create(0))
W = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH
for name, fl in (("kernels only", W), ("lean delivery", W | kmlib.KM_RUN_DELIVER | kmlib.KM_DELIVER_LEAN)):
    for q in range(4):
        bs[q].run(fl, sts[q])
    for q in range(4):
        bs[q].sync()
    # host time of run() when nothing has to be waited for: 8 launches back to back
    t0 = time.perf_counter()
    for i in range(8):
        bs[i % 4].run(fl, sts[i % 4])
    t_run = (time.perf_counter() - t0) / 8
    for q in range(4):
        bs[q].sync()
    # steady state
    N = 80
    t_wait = t_launch = 0.0
    t0 = time.perf_counter()
    for i in range(N):
        q = i % 4
        a = time.perf_counter()
        if fl & kmlib.KM_RUN_DELIVER:
            bs[q].wait_result()
        else:
            bs[q].sync() if i >= 4 and False else None
        c = time.perf_counter()
        bs[q].run(fl, sts[q])
        d = time.perf_counter()
        t_wait += c - a
        t_launch += d - c
    for q in range(4):
        bs[q].wait_result() if fl & kmlib.KM_RUN_DELIVER else bs[q].sync()
    dt = (time.perf_counter() - t0) / N
    print("%-14s: run() alone %.1f us; steady state %.1f us/step = wait %.1f + launch %.1f (host)" %
          (name, t_run * 1e6, dt * 1e6, t_wait / N * 1e6, t_launch / N * 1e6), flush=True)
    t0 = time.perf_counter()
    kmlib.pump(bs, sts, N, fl)
    print("%-14s: the same loop inside the library (km_batch_pump): %.1f us/step" % (name, (time.perf_counter() - t0) / N * 1e6), flush=True)
for b in bs:
    b.close()
db.close()
