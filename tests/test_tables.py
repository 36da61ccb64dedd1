"""Host logic: the tolerant reader for the reference's tables (data files)."""
import os

import numpy as np

from conftest import DATA, load_gv, load_hyper, load_maximin, load_qian
from ccgp_amd.tables import read_table, write_table


def test_qian_shapes():
    D, y, Dt, yt = load_qian()
    assert D.shape == (64, 4) and y.shape == (64,)
    assert Dt.shape == (14, 4) and yt.shape == (14,)
    assert D.min() >= 0.0 and D.max() <= 1.0
    assert 9.5 < y.min() < 9.6 and 35.0 < y.max() < 35.1


def test_hyperpars_shapes():
    H = load_hyper("hx")
    assert H.shape == (624, 4)
    assert H[0].tolist() == [3, 1, 3, 20]
    A = load_hyper("adv")
    assert A.shape == (60, 4)
    assert A[0].tolist() == [3, 1, 5, 8]


def test_bare_tab_separated_without_trailing_newline():
    D14 = load_maximin(14)
    D100 = load_maximin(100)
    assert D14.shape == (14, 2) and D100.shape == (100, 2)
    assert D14.min() >= 0.0 and D14.max() <= 1.0
    assert D100.min() >= -1.0 and D100.max() <= 1.0 and D100.min() < 0


def test_gv_sets():
    for size, ntest in ((50, 150), (90, 110)):
        D, y, Dt, yt = load_gv(size)
        assert D.shape == (size, 9) and Dt.shape == (ntest, 9)


def test_write_table_roundtrip(tmp_path):
    a = np.array([[1.5, 2.0], [3.25, -4.0]])
    p = os.path.join(tmp_path, "t.txt")
    write_table(p, a, ["u", "v"])
    names, b = read_table(p)
    assert names == ["u", "v"]
    np.testing.assert_array_equal(a, b)
    with open(p, "rb") as fh:
        assert fh.read().count(b"\r\n") == 3


def test_results_table_of_the_reference_round_trips_byte_for_byte(tmp_path):
    """Ground Vibrations Emulator/Results/Size 50 Results 1.txt (write.table(as.matrix(Comp.obj)), GV:760-761,
    the only output the reference records): read it, write it back, same bytes -- so a results table
    produced here is diff-able against the reference's."""
    src = os.path.join(DATA, "gv", "results_50_1.txt")
    names, a, rows = read_table(src, with_row_names=True)
    assert a.shape == (150, len(names)) and names[:3] == ["slope", "angle", "top.layer3"]
    assert rows[:5] == ["1", "2", "3", "4", "7"]                  # the test set is a subset of a larger frame
    assert "y.hat.Combined" in names and "y.true" in names
    out = os.path.join(tmp_path, "back.txt")
    write_table(out, a, names, rows)
    with open(src, "rb") as f1, open(out, "rb") as f2:
        assert f1.read() == f2.read()
    # the summary the recorded run implies: RMSPE 2.72, 95 % coverage 0.97
    yt, yh = a[:, names.index("y.true")], a[:, names.index("y.hat.Combined")]
    lo, hi = a[:, names.index("LL.Combined")], a[:, names.index("UL.Combined")]
    assert abs(np.sqrt(np.mean((yt - yh) ** 2)) - 2.72) < 0.01
    assert abs(np.mean((yt >= lo) & (yt <= hi)) - 0.973) < 0.005
