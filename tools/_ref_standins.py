"""Stand-ins that let the REFERENCE's own modules import in the build container (see tools/make_goldens.py for the
rationale): ``pywt`` (16 constants + dwt_max_level, fed from tests/golden/pywt_db2.json which was dumped from a real
PyWavelets 1.1.1) and ``pyevtk`` (no-op VTK writer).  Used only by the golden generators under tools/."""
import json
import os
import sys
import types

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.normpath(os.path.join(HERE, '..', 'tests', 'golden'))


def install():
    if not os.path.isdir(REF):
        sys.exit('the golden generators need the reference checkout at /root/reference (build container only)')
    with open(os.path.join(GOLD, 'pywt_db2.json')) as f:
        table = json.load(f)

    class _Wavelet:
        def __init__(self, name):
            assert name == 'db2', name
            self.name = name
            self.filter_bank = tuple(list(x) for x in table['filter_bank'])
            self.dec_len = table['dec_len']

    def _dwt_max_level(data_len, filter_len):
        flen = filter_len.dec_len if isinstance(filter_len, _Wavelet) else int(filter_len)
        assert flen == 4
        return table['dwt_max_level_flen4'][str(int(data_len))]

    pywt_mod = types.ModuleType('pywt')
    pywt_mod.Wavelet = _Wavelet
    pywt_mod.dwt_max_level = _dwt_max_level
    sys.modules['pywt'] = pywt_mod
    pyevtk_mod = types.ModuleType('pyevtk')
    pyevtk_hl = types.ModuleType('pyevtk.hl')
    pyevtk_hl.imageToVTK = lambda *a, **k: None
    pyevtk_mod.hl = pyevtk_hl
    sys.modules['pyevtk'] = pyevtk_mod
    sys.modules['pyevtk.hl'] = pyevtk_hl
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return table
