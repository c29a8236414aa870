"""The library's lineitem generator kernel against its host twin, and TPC-H Q1 over a generated table against the oracle.
bench.py performs the same spot check before every headline measurement (and falls back to the torch generators if it
fails), so a defect here cannot corrupt a measurement; the kernel was written after this round's GPU budget was spent,
hence the non-strict xfail: a pass shows up as XPASS, a failure does not mask the rest of the suite."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.xfail(strict=False, reason="generator kernel not yet validated on a GPU (added after this round's GPU budget was spent)")]


@pytest.mark.parametrize("n", [0, 1, 4097, 1_000_003])
def test_device_generator_equals_host_twin(pl, n):
    from polars_amd import datagen
    df = datagen.lineitem_native(pl, n, seed=12)
    assert df.height == n and df.columns == datagen.LINEITEM_Q1_COLS
    want = datagen.lineitem_native_host(0, n, 12)
    for c in datagen.LINEITEM_Q1_COLS:
        assert np.array_equal(df[c].to_numpy(), want[c]), c


def test_q1_over_generated_table_matches_oracle(pl, orc):
    from polars_amd import datagen, queries
    n = 500_000
    df = datagen.lineitem_native(pl, n, seed=3)
    g = queries.q1(df.lazy()).collect().sort_host(["l_returnflag", "l_linestatus"])
    assert "fused_scan[aot]" in pl.last_plan(), pl.last_plan()
    want = orc.q1(datagen.lineitem_native_host(0, n, 3), datagen.us(1998, 9, 2))
    assert [datagen.FLAGS.index(x) for x in g["l_returnflag"]] == want["l_returnflag"].tolist()
    assert g["count_order"] == want["count_order"].tolist() and g["sum_qty"] == want["sum_qty"].tolist()
    for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
        assert np.allclose(np.array(g[c]), want[c], rtol=1e-6, atol=0), c
