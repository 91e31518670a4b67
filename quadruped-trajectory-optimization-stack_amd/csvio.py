"""Plan CSV: 37 columns, no header, 1 kHz (producer: the solver's build/traj.csv; consumers
scripts/run.py:129-137,184-188, QTOS/utils.py:107-148, QTOS/combiner.py:263-274).
Values are printed like the reference's C++ stream does (default precision 6 = ``%g``)."""
import os

import numpy as np

COLS = 37
COLUMN_MAP = {  # QTOS/utils.py:107-148 vec_to_cmd_pose (indices into a full row incl. time)
    "t": slice(0, 1), "com": slice(1, 4), "euler": slice(4, 7), "FL": slice(7, 10),
    "FR": slice(10, 13), "HL": slice(13, 16), "HR": slice(16, 19), "com_vel": slice(19, 22),
    "euler_rate": slice(22, 25), "FL_force": slice(25, 28), "FR_force": slice(28, 31),
    "HL_force": slice(31, 34), "HR_force": slice(34, 37),
}


def write_csv(path, rows, n_threads=0):
    """rows [n, 37] -> the text file, through the library's native writer (qtos_write_csv, csrc/csv_writer.hpp: "%g" like the
    solver's C++ stream, rows formatted by a few threads).  A 5001-row plan takes 2 ms instead of the 26 ms of the Python loop
    below -- which was 93 % of the time of one plan from its flags to the file."""
    import ctypes as C
    from . import capi
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    assert rows.ndim == 2 and rows.shape[1] == COLS
    rc = capi.load().qtos_write_csv(os.fsencode(str(path)), rows.ctypes.data_as(C.POINTER(C.c_double)), int(rows.shape[0]), int(n_threads))
    if rc != 0:
        raise OSError("qtos_write_csv(%r) failed with %d (-2: cannot open the file, -3: short write)" % (str(path), rc))


def write_csv_python(path, rows):
    """The same file by Python's own "%g" (the statement of the format the native writer is held to: tests/test_boundary.py)."""
    rows = np.asarray(rows)
    assert rows.ndim == 2 and rows.shape[1] == COLS
    with open(path, "w") as f:
        for r in rows:
            f.write(",".join("%g" % v for v in r) + "\n")


def read_csv(path):
    return np.loadtxt(path, delimiter=",", ndmin=2)
