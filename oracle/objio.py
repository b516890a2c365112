"""TEST INFRASTRUCTURE ONLY.  Minimal Wavefront-OBJ point reader used to feed the checkers.

Semantics follow what the reference's loader yields (source/common/loader.cpp:30-67): assimp is run with
aiProcess_Triangulate and WITHOUT aiProcess_JoinIdenticalVertices, and the loader copies mesh->mVertices: the OBJ importer makes
one vertex per face-vertex reference and the triangulation step only re-indexes them, so a mesh contributes one point per corner
of every face AS WRITTEN, in face order (bunny.obj: 4 968 triangles -> 14 904 points; bird.obj: 8 752 quads -> 35 008 points,
cf. source/common/testset.cpp:22-26 -- round 5: quads used to be fan-triangulated here into 6 points).
A file without faces falls back to its bare vertex list.
"""
import numpy as np


def load_obj_points(path):
    verts = []
    corners = []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                ids = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    ids.append(i - 1 if i > 0 else len(verts) + i)
                corners.extend(ids)
    v = np.asarray(verts, dtype=np.float32).reshape(-1, 3)
    if not corners:
        return v
    return np.ascontiguousarray(v[np.asarray(corners, dtype=np.int64)])
