"""Writes profiles/rNN_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each in its own run, as
MI355X_MICROARCH.md prescribes) of `python bench.py`: HBM-side bytes per launch of the dominant kernel, with the
gfx950 correction (FETCH_SIZE counts 128-B requests as 64 B: x2; both counters are in KiB), stamped with the digest of
the kernel sources so that bench.py only reports a traffic figure measured on the sources in the tree.
    python tools/traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring> <precision> <out.json>"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_digest


def mean_counter(path, pat, name):
    rows = [r for r in csv.DictReader(open(path)) if pat in r['Kernel_Name'] and r['Counter_Name'] == name]
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows]
    keep = [r for r, d in zip(rows, dur) if d > 0.7 * max(dur)]
    return sum(float(r['Counter_Value']) for r in keep) / len(keep), len(keep)


fetch, nf = mean_counter(sys.argv[1], sys.argv[3], 'FETCH_SIZE')
write, nw = mean_counter(sys.argv[2], sys.argv[3], 'WRITE_SIZE')
out = {}
if os.path.exists(sys.argv[5]):
    out = json.load(open(sys.argv[5]))
out[sys.argv[4]] = {'kernel': sys.argv[3], 'bytes_per_launch': int(2 * fetch * 1024 + write * 1024), 'fetch_kib_raw': fetch,
                    'write_kib': write, 'dispatches': [nf, nw], 'sources': kernel_sources_digest(),
                    'note': 'FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE as is; KiB -> bytes'}
json.dump(out, open(sys.argv[5], 'w'), indent=1)
print(json.dumps(out[sys.argv[4]]))
