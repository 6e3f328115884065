#!/usr/bin/env python3
"""Performance floors of the round, kept OUT of the correctness suite (a busy or slower box must not fail `pytest -m gpu`):

    python bench.py > gpurun_out/bench.json && python tools/perf_floor.py gpurun_out/bench.json

Reads the bench line's compact `configs` list and checks each configuration against the floor below (about 7 % under what
the round measured on the pool's boxes).  Exit code 1 and a table when one is missed."""
import json
import sys

FLOORS = {   # cfg substring -> (key, floor)
    "4096x20x10 2D raw (timed region)": ("frac", 0.50),
    "configs[2] MAAC-R, PMI hidden 128": ("G", 11.6),
    "configs[2] MAAC-R, PMI hidden 64": ("G", 15.9),
    "configs[2] MAAC-R dense": ("G", 5.9),
    "configs[3] 3-D": ("frac", 0.315),
    "chip-filling": ("frac", 0.64),
}


def main(path):
    line = json.loads([ln for ln in open(path).read().splitlines() if ln.strip().startswith("{")][-1])
    bad = 0
    for c in line["configs"]:
        for sub, (key, floor) in FLOORS.items():
            if sub in c["cfg"] and key in c:
                ok = c[key] >= floor
                bad += not ok
                print(f"{'ok  ' if ok else 'MISS'} {c['cfg']:<70s} {key} = {c[key]:<8g} floor {floor}")
    cl = [c for c in line["configs"] if c["cfg"].startswith("closed loop")]
    if cl:
        print("closed loop:", {k: v for k, v in cl[0].items() if k != "cfg"})
        for key, floor in (("actor_fused", 11.9), ("actor_chunks", 9.2), ("actor_fused_maac_r", 6.9), ("greedy_fused", 15.2), ("greedy_chunks", 12.6)):
            if key in cl[0]:
                ok = cl[0][key] >= floor
                bad += not ok
                print(f"{'ok  ' if ok else 'MISS'} closed loop {key:<56s} G = {cl[0][key]:<8g} floor {floor}")
    cp = [c for c in line["configs"] if c["cfg"].startswith("compat")]
    if cp:      # the B = 1 adapter: wall microseconds per env.step, ceilings (about 25 % over what the round measured: host-side, noisy)
        for key, ceil in (("20x10 MAAC", 45.0), ("20x10 MAAC-R H=128", 58.0), ("50x25 MAAC", 65.0)):
            if key in cp[0]:
                ok = cp[0][key][0] <= ceil
                bad += not ok
                print(f"{'ok  ' if ok else 'MISS'} compat {key:<61s} us = {cp[0][key][0]:<8g} ceiling {ceil}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
