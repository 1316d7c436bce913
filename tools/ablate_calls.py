#!/usr/bin/env python3
"""TIMING-ONLY ablation of whole libsfamd entry points inside the training step (results are garbage).

    SF_SKIP_CALLS=sf_bn_finalize,sf_bn_bwd_finalize python tools/ablate_calls.py --preset SLOWFAST_8x8_R50 --steps 10 \
        --no-kernel-profile --no-cpu-baseline --no-secondary

Every native call whose name is listed in SF_SKIP_CALLS is NOT issued (its outputs keep whatever the allocator handed out); the
rest of bench.py runs unchanged, so `ms_per_step` of the printed line against an un-ablated run is what those launches cost
inside the replayed graph -- the upper bound of what folding them into their producers can return.  Diagnostic tool: lives
outside the package, the product library has no such switch.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    skip = {s for s in os.environ.get("SF_SKIP_CALLS", "").split(",") if s}
    from slowfast_amd import lib
    counts = {}

    def observer(name, thunk, work):
        if name in skip:
            counts[name] = counts.get(name, 0) + 1
            return 0
        return thunk()

    if skip:
        lib.set_call_observer(observer)
    import bench
    assert "--no-kernel-profile" in sys.argv or not skip, "the kernel profiler installs its own observer: pass --no-kernel-profile"
    bench.main()
    if skip:
        print("ablate_calls: skipped", counts, file=sys.stderr)


if __name__ == "__main__":
    main()
