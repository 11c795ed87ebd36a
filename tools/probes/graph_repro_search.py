#!/usr/bin/env python3
"""Exhaustive search for the SHORTEST event graph that sends hipStreamEndCapture into unbounded recursion (tools/probes/graph_fork_join_repro.hip,
pattern 98): all edge sequences over three streams up to the given length, each in its own process.  usage: graph_repro_search.py [maxlen=4]"""
import itertools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "tools", "probes", "graph_fork_join_repro.bin")
maxlen = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
edges = [(a, b) for a in range(NS) for b in range(NS) if a != b]
found = []
for n in range(1, maxlen + 1):
    tried = 0
    for seq in itertools.product(edges, repeat=n):
        cap = {0}
        ok = True
        for a, b in seq:
            if a not in cap:
                ok = False
                break
            cap.add(b)
        if not ok:
            continue
        # skip sequences that extend an already known crashing prefix pattern (report minimal ones only)
        txt = " ".join(f"{a}{b}" for a, b in seq)
        if any(f in txt for f in found):
            continue
        tried += 1
        r = subprocess.run([BIN, "98", "0", "0", "0", txt], capture_output=True, text=True)
        if r.returncode != 0:
            print(f"CRASH rc={r.returncode} len={n}: {txt}   {r.stdout.strip()[:100]}", flush=True)
            found.append(txt)
    print(f"length {n}: {tried} sequences tried, {len(found)} minimal crashing so far", flush=True)
    if found:
        break
