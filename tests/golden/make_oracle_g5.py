'''
G5 fixture capture -- ORACLE-GENERATED, NOT REFERENCE-GENERATED.

Writes tests/golden/oracle_g5_<case>.npz from the live oracle (oracle/g5.py lists the
cases and what each file holds).  Regenerate ONLY when an oracle change is intended and
reviewed against the reference lines it cites; the CPU test
tests/test_oracle_cpu.py::test_g5_live_oracle_matches_committed_fixtures then pins the
new state.

Usage:  python tests/golden/make_oracle_g5.py
'''
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import g5  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    for name in sorted(g5.CASES):
        packed = g5.pack(g5.case_outputs(name))
        packed['_label'] = np.array('oracle-generated (oracle/danet_oracle.py + oracle/torch_ref.py, '
                                    'float64); NOT produced by the reference')
        path = os.path.join(OUT, 'oracle_g5_%s.npz' % name)
        np.savez_compressed(path, **packed)
        print('%s: %d arrays, %.1f KB' % (path, len(packed), os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    main()
