"""A small two-table lookup system for the twin of /root/reference/src/logup/: a `provider` trace (a preprocessed table of
(index, value) rows provided to two requiring traces) and a `requirer` trace whose rows each look two entries up, one of them
behind an is_real flag.  Built so that the LogUp sums of the tables cancel when the multiplicities are the true counts."""
import numpy as np

from oracle import logup as ol
from oracle import stark as os_

P = os_.P
Z, R, GAMMA = (3, 1, 4, 1), (5, 9, 2, 6), (5, 3, 5, 8)


def system(height=16, seed=1):
    rng = np.random.default_rng(seed)
    table = [(i, int(rng.integers(0, P))) for i in range(height)]
    # provider: preprocessed = [index, value]; main = [is_real] ; provides (index, value) -- identity column = row index
    prov_prep = np.array(table, dtype=np.uint64)
    prov_main = np.ones((height, 1), dtype=np.uint64)
    provides = [([([(ol.PREP, 0, 1)], 0), ([(ol.PREP, 1, 1)], 0)], ([(ol.MAIN, 0, 1)], 0))]
    # requirer (trace index 0): main = [i1, v1, i2, v2, flag2]; requires (i1, v1) always and (i2, 2 * v2 / 2 + 0) when flag2
    req_main = np.zeros((height, 5), dtype=np.uint64)
    counts = np.zeros((height, 1), dtype=np.uint32)
    for r_ in range(height):
        i1, i2 = int(rng.integers(0, height)), int(rng.integers(0, height))
        flag = int(rng.integers(0, 2))
        req_main[r_] = [i1, table[i1][1], i2, table[i2][1], flag]
        counts[i1, 0] += 1
        if flag:
            counts[i2, 0] += 1
    requires = [([([(ol.MAIN, 0, 1)], 0), ([(ol.MAIN, 1, 1)], 0)], None),
                ([([(ol.MAIN, 2, 1)], 0), ([(ol.MAIN, 3, 3), (ol.MAIN, 3, P - 2)], 0)], ([(ol.MAIN, 4, 1)], 0))]  # 3 v - 2 v = v: a two-term form
    identity = np.arange(height, dtype=np.uint64)
    return {"height": height, "identity": identity, "prov_prep": prov_prep, "prov_main": prov_main, "provides": provides, "req_main": req_main,
            "requires": requires, "multiplicities": [([0], counts.tolist())]}


def ef_rows(rows):
    return [[list(c) for c in r] for r in rows]
