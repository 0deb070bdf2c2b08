#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference holds no numeric golden vectors for the lookup path (its CI data is on an NVIDIA-internal
volume, /root/reference/.gitlab-ci.yml:70-72,87) and its engine cannot be built or imported here, so these
fixtures pin the STRUCTURAL facts the reference does state, evaluated with the NumPy restatement in
oracle/hps_oracle.py (sort + searchsorted, no hashing):
  wdl        W&D request of the deployment sample: tables D=[1,16], 10 samples, keys/sample [2,26] -> OUTPUT0
             shape [4180], NumSample 10 (samples/Hierarchical_Parameter_Server_Deployment.ipynb:738-747,793-795)
  identity   table files written with the struct.pack('q') / struct.pack('f') recipe of
             samples/hps-triton-ensemble/01_model_training.ipynb:498-504; keys 0..R-1 => lookup(k) == row k
  default    keys absent from every tier return default_value_for_each_table (1.0 as in 02_...ipynb:220,
             0.0 as in README.md:150)   docs/hierarchical_parameter_server.md:244-246
  dups       duplicate keys each get a full copy, output order = input order (docs/architecture.md:308-318)
  tf3072     single table D=16, 1024 samples x 3 keys, NUMKEYS [[3072]] (02_...ipynb:661-662)
"""
import struct
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import hps_oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent
rng = np.random.default_rng(20260929)
g = {}

# ---- wdl -------------------------------------------------------------------------------------------
k0 = rng.permutation(400)[:64].astype(np.int64)
k1 = rng.permutation(400)[:48].astype(np.int64)
r0 = O.np_synth_rows(O.SEED, 0, k0, 1)
r1 = O.np_synth_rows(O.SEED, 1, k1, 16)
batch = 10
q = np.concatenate([rng.choice(k0, batch * 2), rng.choice(k1, batch * 26)]).astype(np.int64)
q[[3, 17, 40, 200]] = [1000, 1001, 1002, 1003]  # four keys that exist nowhere
g.update(wdl_k0=k0, wdl_r0=r0, wdl_k1=k1, wdl_r1=r1, wdl_keys=q, wdl_numkeys=np.array([batch * 2, batch * 26], np.int32),
         wdl_defaults=np.array([0.0, 0.0], np.float32),
         wdl_expected=O.np_lookup([(k0, r0), (k1, r1)], q, [batch * 2, batch * 26], [0.0, 0.0]))
assert g["wdl_expected"].shape == (4180,)

# ---- identity table written with the notebook's struct.pack recipe -----------------------------------
R, D = 32, 4
ident_rows = O.np_synth_rows(O.SEED, 7, np.arange(R), D)
d = OUT / "identity_table"
d.mkdir(exist_ok=True)
with open(d / "key", "wb") as fk, open(d / "emb_vector", "wb") as fv:
    for k in range(R):
        fk.write(struct.pack("q", k))
        fv.write(struct.pack(str(D) + "f", *ident_rows[k].tolist()))
g.update(identity_rows=ident_rows)

# ---- default fill -----------------------------------------------------------------------------------
qd = np.array([0, 31, 32, 33, 10**12, -1, 5], np.int64)
g.update(default_keys=qd,
         default_expected_1=O.np_lookup([(np.arange(R, dtype=np.int64), ident_rows)], qd, [qd.size], [1.0]),
         default_expected_0=O.np_lookup([(np.arange(R, dtype=np.int64), ident_rows)], qd, [qd.size], [0.0]))

# ---- duplicates + order ------------------------------------------------------------------------------
qq = np.array([5, 5, 3, 5, 31, 3, 0, 0, 0], np.int64)
g.update(dups_keys=qq, dups_expected=O.np_lookup([(np.arange(R, dtype=np.int64), ident_rows)], qq, [qq.size], [0.0]))

# ---- duplicate keys inside a table file: the last row wins --------------------------------------------
fk = np.array([9, 4, 9, 7, 4], np.int64)
fr = O.np_synth_rows(O.SEED, 9, np.arange(5), 2)
g.update(filedup_keys=fk, filedup_rows=fr, filedup_query=np.array([9, 4, 7, 1], np.int64),
         filedup_expected=O.np_lookup([(fk, fr)], [9, 4, 7, 1], [4], [2.5]))

# ---- tf ensemble shape -------------------------------------------------------------------------------
kt = rng.permutation(1000)[:200].astype(np.int64)
rt = O.np_synth_rows(O.SEED, 3, kt, 16)
qt = rng.choice(np.concatenate([kt, [5000, 5001]]), 3072).astype(np.int64)
g.update(tf_k=kt, tf_r=rt, tf_keys=qt, tf_expected=O.np_lookup([(kt, rt)], qt, [3072], [1.0]))
assert g["tf_expected"].shape == (3072 * 16,)

# ---- raw generator KAT: first elements of the synthetic-table recipe -------------------
g.update(synth_t3_k5_d16=O.np_synth_rows(O.SEED, 3, np.arange(5, 9), 16), mix64_of_0_1_2=O.np_mix64(np.arange(3, dtype=np.uint64)))

np.savez_compressed(OUT / "hps_golden.npz", **g)
print("wrote", OUT / "hps_golden.npz", sum(v.nbytes for v in g.values()), "bytes raw")
