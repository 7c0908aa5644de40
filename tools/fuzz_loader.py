"""Mutation fuzzing of the model loaders (host code only): python tools/fuzz_loader.py model.glb|model.obj N SEED.
Every mutated file must either load or raise a clean error; the index of the file being tried goes to stderr, so a
crash names its case. 6000 mutations of the golden models and of a small OBJ/MTL pair: no crash."""
import sys, os, random, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'path-tracer_amd')); sys.path.insert(0, ROOT)
import torch, ptamd
src = sys.argv[1]; n = int(sys.argv[2]); seed = int(sys.argv[3])
data = bytearray(open(src, 'rb').read())
rnd = random.Random(seed)
ok = err = 0
tmp = tempfile.mktemp(suffix=os.path.splitext(src)[1])
for i in range(n):
    d = bytearray(data)
    mode = rnd.randrange(4)
    if mode == 0:
        for _ in range(rnd.randrange(1, 8)):
            d[rnd.randrange(len(d))] = rnd.randrange(256)
    elif mode == 1:
        d = d[:rnd.randrange(1, len(d))]
    elif mode == 2:  # mutate inside the JSON chunk (first 4 KB)
        for _ in range(rnd.randrange(1, 6)):
            d[rnd.randrange(12, min(len(d), 4096))] = rnd.choice(b'0123456789-{}[],:"e.')
    else:
        p = rnd.randrange(len(d)); d[p:p] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 64)))
    open(tmp, 'wb').write(d)
    print(i, mode, flush=True, file=sys.stderr)
    try:
        ptamd.load_model(tmp, node_transforms=bool(i & 1), allow_obj=tmp.endswith(".obj")); ok += 1
    except Exception as e:
        err += 1
print("ok", ok, "err", err)
