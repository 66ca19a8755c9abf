#!/usr/bin/env python
"""Acceptance check a maintainer with weights and videos runs first: a produced `*_frameIdx.json` against the published one
(evaluation/jsons_idx/TSPO_<dataset>_frameIdx.json of the reference; written by mp_tools/change_score_tch.py:22-44, read by
lmms-eval llava_vid_tspo.py:362-380).

    python tools/compare_frame_idx.py produced.json published.json [--key id|question_id] [--worst 10]

Per doc (joined on the dataset's key): exact match of the frame list, Jaccard overlap of the frame sets, and - because the
selection is a top-k over near-tied scores - the overlap when a produced frame may sit one candidate step away from a
published one.  Prints the summary as one JSON line; exit code 1 if any doc is missing from either side."""
from __future__ import annotations

import argparse
import json
import math
import sys


def _key_of(docs, key):
    if key:
        return key
    for k in ("question_id", "id", "index"):
        if all(k in d for d in docs):
            return k
    raise SystemExit("no common join key (question_id / id / index): pass --key")


def _step(frames):
    s = 0
    for v in frames:
        s = math.gcd(s, int(v))
    return max(s, 1)


def compare(produced, published, key=None, worst=10):
    key = _key_of(published, key)
    pub = {d[key]: d for d in published if "frame_idx" in d}
    pro = {d[key]: d for d in produced if "frame_idx" in d}
    rows, missing = [], sorted(set(pub) - set(pro), key=str)
    for k, d in pub.items():
        if k not in pro:
            continue
        a, b = [float(x) for x in pro[k]["frame_idx"]], [float(x) for x in d["frame_idx"]]
        sa, sb = set(a), set(b)
        st = _step(b)
        near = sum(1 for x in sa if x in sb or x - st in sb or x + st in sb)
        rows.append({"key": k, "exact": a == b, "jaccard": len(sa & sb) / max(1, len(sa | sb)), "n_produced": len(a),
                     "n_published": len(b), "within_one_step": near / max(1, len(sa)), "sorted": a == sorted(a)})
    n = len(rows)
    summary = {
        "join_key": key, "docs_published": len(pub), "docs_produced": len(pro), "docs_compared": n,
        "missing_in_produced": len(missing), "extra_in_produced": len(set(pro) - set(pub)),
        "exact_match_rate": sum(r["exact"] for r in rows) / max(1, n),
        "mean_jaccard": sum(r["jaccard"] for r in rows) / max(1, n),
        "min_jaccard": min((r["jaccard"] for r in rows), default=1.0),
        "mean_within_one_step": sum(r["within_one_step"] for r in rows) / max(1, n),
        "length_mismatches": sum(r["n_produced"] != r["n_published"] for r in rows),
        "unsorted_produced": sum(not r["sorted"] for r in rows),
        "worst": sorted(rows, key=lambda r: r["jaccard"])[:worst],
        "missing_keys": missing[:worst],
    }
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("produced")
    ap.add_argument("published")
    ap.add_argument("--key", default=None)
    ap.add_argument("--worst", type=int, default=10)
    a = ap.parse_args(argv)
    s = compare(json.load(open(a.produced)), json.load(open(a.published)), a.key, a.worst)
    print(json.dumps(s))
    return 1 if (s["missing_in_produced"] or s["extra_in_produced"]) else 0


if __name__ == "__main__":
    sys.exit(main())
