"""Dump the db2 filter bank and dwt_max_level table from a REAL PyWavelets install.

Run in the build container with the interpreter that has pywt (no torch there):
    /opt/conda/bin/python3.9 tools/dump_pywt_constants.py
Writes tests/golden/pywt_db2.json.  The torch-bearing interpreter has no pywt, so this JSON is the
only channel through which PyWavelets' numbers reach the oracle tests and tools/make_goldens.py.
"""
import json
import os
import pywt

w = pywt.Wavelet('db2')
out = {
    'pywt_version': pywt.__version__,
    'name': w.name,
    'filter_bank': [list(map(float, f)) for f in w.filter_bank],   # dec_lo, dec_hi, rec_lo, rec_hi
    'dec_len': w.dec_len,
    'dwt_max_level_flen4': {str(n): int(pywt.dwt_max_level(n, 4)) for n in range(1, 300)},
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'pywt_db2.json')
with open(path, 'w') as f:
    json.dump(out, f, indent=1)
print('wrote', os.path.normpath(path))
