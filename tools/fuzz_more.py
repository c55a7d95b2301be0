import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import tests.test_gpu_parity as t
bad = []
for seed in range(40, 400):
    try:
        t.test_randomised_schemes_partitions_and_tables(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200])); print("FAIL", seed, str(e)[:200], flush=True)
    if seed % 40 == 0: print("seed", seed, flush=True)
for seed in range(12, 120):
    try:
        t.test_ng21_randomised_expanded_schemes(seed)
    except AssertionError as e:
        bad.append(("ng21", seed, str(e)[:200])); print("FAIL ng21", seed, flush=True)
for name in ("test_exact_search_randomised_layouts_and_tables",):
    fn = getattr(t, name, None)
    if fn is None: continue
    for seed in range(100, 700):
        try:
            fn(seed)
        except AssertionError as e:
            bad.append((name, seed, str(e)[:200])); print("FAIL", name, seed, flush=True)
        except Exception as e:
            bad.append((name, seed, repr(e)[:200])); print("ERR", name, seed, repr(e)[:200], flush=True)
        if seed % 50 == 0: print(name, seed, flush=True)
print("done, failures:", bad)
