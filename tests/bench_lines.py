"""What bench.py puts on stdout: the FULL record on an earlier line ("[bench] full record: {...}", also written to bench_extras.json) and the
headline line -- the one the driver parses -- LAST.  Helpers for the tests that read either."""
import json

FULL_PREFIX = "[bench] full record: "
HEADLINE_LIMIT = 4096


def headline_lines(stdout: str):
    return [ln for ln in stdout.splitlines() if ln.startswith("{")]


def full_record(stdout: str):
    recs = [ln[len(FULL_PREFIX):] for ln in stdout.splitlines() if ln.startswith(FULL_PREFIX)]
    assert len(recs) == 1, stdout[-2000:]
    return json.loads(recs[0])


def split(stdout: str):
    """-> (headline, full).  Asserts the contract: exactly one headline line, it is the LAST non-empty stdout line, below 4 KB, and the full record precedes it."""
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    heads = headline_lines(stdout)
    assert len(heads) == 1, stdout[-2000:]
    assert lines[-1] == heads[0], lines[-1][:300]
    assert len(heads[0].encode()) < HEADLINE_LIMIT, len(heads[0])
    return json.loads(heads[0]), full_record(stdout)
