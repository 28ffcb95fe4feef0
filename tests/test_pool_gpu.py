"""The context's device-memory pool (lurk_amd/csrc/ctx.hip: pool_alloc / pool_release): accounting, and the trim-and-retry
path taken when the driver is out of memory, forced through lurkhip_debug_inject_alloc_failures (that path
freed the selector tables twice and left a sticky HIP error behind)."""
import numpy as np
import pytest

import lurk_amd
from lurk_amd import _native as N
from lurk_amd import commit as lcommit
from lurk_amd import synth

pytestmark = pytest.mark.gpu


def _mats(seed, log_n=12):
    return [synth.field_elements((1 << log_n, 40), seed=seed), synth.field_elements((1 << (log_n - 3), 9), seed=seed + 1)]


def test_pool_accounting_and_forced_oom_retry():
    with lurk_amd.Context(0) as ctx:
        c = lcommit.commit(ctx, _mats(1), log_blowup=1)
        want = [int(v) for v in c.root]
        c.close()
        st = ctx.pool_stats()
        assert st["cached_bytes"] > 0 and st["peak_bytes"] >= st["cached_bytes"] + st["live_bytes"] and st["oom_retries"] == 0
        mallocs = st["mallocs"]
        # same shapes again: served from the free lists, no hipMalloc
        c = lcommit.commit(ctx, _mats(1), log_blowup=1)
        assert [int(v) for v in c.root] == want
        c.close()
        assert ctx.pool_stats()["mallocs"] == mallocs
        # a new block size with the driver "out of memory" once: cached blocks go back to the driver, the retry succeeds, nothing
        # sticks to the stream (the next commitment runs and gives the same root)
        ctx.debug_inject_alloc_failures(1)
        c2 = lcommit.commit(ctx, _mats(7, log_n=13), log_blowup=1)
        c2.close()
        st = ctx.pool_stats()
        assert st["oom_retries"] == 1
        c = lcommit.commit(ctx, _mats(1), log_blowup=1)
        assert [int(v) for v in c.root] == want
        c.close()
        # both attempts fail: a clean LURKHIP_ERR_OOM, and the context stays usable
        ctx.pool_trim()
        ctx.debug_inject_alloc_failures(1000)
        with pytest.raises(lurk_amd.LurkHipError) as e:
            lcommit.commit(ctx, _mats(1), log_blowup=1)
        assert e.value.status == N.ERR_OOM
        ctx.debug_inject_alloc_failures(0)
        c = lcommit.commit(ctx, _mats(1), log_blowup=1)
        assert [int(v) for v in c.root] == want
        c.close()
        ctx.pool_reset_peak()
        st = ctx.pool_stats()
        assert st["peak_bytes"] == st["live_bytes"] + st["cached_bytes"]
