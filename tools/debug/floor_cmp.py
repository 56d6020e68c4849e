import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    if not np.array_equal(a[k], b[k]):
        bad = np.argwhere(a[k] != b[k])
        print(k, a[k].shape, "first diffs", bad[:6].tolist())
        r = bad[0][0]
        print(" a", a[k][r].tolist() if a[k].ndim > 1 else a[k].tolist())
        print(" b", b[k][r].tolist() if b[k].ndim > 1 else b[k].tolist())
        break
else:
    print("identical")
