"""Ad-hoc widening of the randomised parity sweeps: the test bodies of tests/test_gpu_random_sweep_hetero.py over
hundreds of further random cases (hetero sampling in its 4 variants, HGT, budget, homogeneous sampling, walks, negatives,
the window-ordered launch, the partitioned sampler) against the oracle.  Run on a GPU box
from the repo root: `python tools/stress_parity.py`."""
import os, sys
ROOT = os.getcwd()
for p in ("oracle", "tests", "tch-geometric_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import tch_geometric as tg
import test_gpu_random_sweep_hetero as T
bad = 0
for case in range(1000, 1400):
    try:
        T.test_hetero_neighbor_sampling_random(tg, case)
    except Exception as e:  # noqa
        bad += 1
        print("FAIL hetero case", case, type(e).__name__, str(e)[:200], flush=True)
        if bad > 5:
            break
for case in range(1000, 1120):
    try:
        T.test_hgt_and_budget_random(tg, case)
    except Exception as e:  # noqa
        bad += 1
        print("FAIL hgt/budget case", case, type(e).__name__, str(e)[:200], flush=True)
        if bad > 10:
            break
import test_gpu_random_sweep as S  # noqa: E402
import test_gpu_random_sweep_windowed as W  # noqa: E402
from tch_geometric import _cabi  # noqa: E402
for name, fn, lo, hi in (("neighbor sampling", S.test_neighbor_sampling_random_cases, 1000, 1300),
                         ("walks / negatives", S.test_walks_and_negatives_random_cases, 1000, 1100),
                         ("windowed launch", W.test_windowed_launch_random_cases, 1000, 1150),
                         ("partitioned", W.test_partitioned_random_cases, 1000, 1100)):
    for case in range(lo, hi):
        try:
            fn(_cabi, case)
        except Exception as e:  # noqa
            bad += 1
            print("FAIL", name, "case", case, type(e).__name__, str(e)[:200], flush=True)
            if bad > 20:
                break
    print(name, "done", flush=True)
print("done, failures:", bad)
