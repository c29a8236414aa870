"""Parquet scan end to end on the GPU: TPC-H Q1 straight from a 16-column lineitem file equals the oracle on the same rows, only
the needed columns / row groups cross PCIe.  The scan planning is pinned on the CPU (tests/test_io_cpu.py)."""
import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

pytestmark = pytest.mark.gpu


def test_q1_from_parquet(pl, orc, tmp_path):
    from polars_amd import datagen, queries
    n = 200_000
    li = datagen.lineitem_host(n, seed=8)
    order = np.argsort(li["l_shipdate"], kind="stable")
    t = pa.table({"l_orderkey": pa.array(np.arange(n)), "l_quantity": pa.array(li["l_quantity"][order]), "l_extendedprice": pa.array(li["l_extendedprice"][order]),
                  "l_discount": pa.array(li["l_discount"][order]), "l_tax": pa.array(li["l_tax"][order]),
                  "l_returnflag": pa.array([datagen.FLAGS[c] for c in li["l_returnflag"][order]], pa.large_string()),
                  "l_linestatus": pa.array([datagen.STATUS[c] for c in li["l_linestatus"][order]], pa.large_string()),
                  "l_shipdate": pa.array(li["l_shipdate"][order], pa.timestamp("us")), "l_comment": pa.array(["x"] * n, pa.large_string())})
    path = str(tmp_path / "lineitem.parquet")
    pq.write_table(t, path, row_group_size=10_000)
    lf = queries.q1(pl.scan_parquet(path))
    out = lf.collect().sort_host(["l_returnflag", "l_linestatus"])
    node = lf._node
    while node.kind != "scan":
        node = node.input
    read = node.frame.last_read
    assert sorted(read["columns"]) == sorted(datagen.LINEITEM_Q1_COLS) and read["row_groups"] < read["of_row_groups"]
    want = orc.q1(li, datagen.us(1998, 9, 2))
    assert [datagen.FLAGS.index(x) for x in out["l_returnflag"]] == want["l_returnflag"].tolist()
    assert [datagen.STATUS.index(x) for x in out["l_linestatus"]] == want["l_linestatus"].tolist()
    assert out["count_order"] == want["count_order"].tolist() and out["sum_qty"] == want["sum_qty"].tolist()
    for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
        assert np.allclose(np.array(out[c]), want[c], rtol=1e-6, atol=0), c
    # eager read of two columns
    df = pl.read_parquet(path, columns=["l_quantity", "l_tax"])
    assert df.columns == ["l_quantity", "l_tax"] and df.height == n and df["l_quantity"].sum() == int(li["l_quantity"].sum())
