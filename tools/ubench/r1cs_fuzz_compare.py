"""r1cs_fuzz_compare.py a.npz b.npz [...] — every array of every file equal to the first's?"""
import sys, numpy as np
ref = np.load(sys.argv[1])
ok = True
for path in sys.argv[2:]:
    other = np.load(path)
    for k in ref.files:
        same = np.array_equal(ref[k], other[k])
        ok = ok and same
        print(f"{path} vs {sys.argv[1]}: {k}: {'equal' if same else 'DIFFERENT at ' + str(np.nonzero(ref[k] != other[k])[0][:10].tolist())} ({len(ref[k])} bodies, {int(np.count_nonzero(ref[k]))} non-zero)")
sys.exit(0 if ok else 1)
