"""Host helpers of the reference's loader that the path and its output step use (ref:bundle_adjust/loader.py:23-40, 74-88, 232-268, 384-406)."""
import sys


def flush_print(*args, **kwargs):
    print(*args, **kwargs)
    sys.stdout.flush()


def display_dict(d):
    """Pretty-print a configuration dict, one `key: value` per line."""
    width = max((len(str(k)) for k in d), default=0)
    for k, v in d.items():
        print("    - {}: {}".format(str(k).ljust(width), v))
    print("")


# ---- output files of the pipeline (ref:bundle_adjust/ba_pipeline.py:379-427, 606-620; ref:bundle_adjust/loader.py:232-238, 384-406)
def save_rpcs(filenames, rpcs):
    """Write RPC models to text files, creating the directories (ref:bundle_adjust/loader.py:232-238)."""
    import os

    for fn, rpc in zip(filenames, rpcs):
        os.makedirs(os.path.dirname(fn) or ".", exist_ok=True)
        rpc.write_to_file(fn)


def write_point_cloud_ply(filename, point_cloud, color=(None, None, None)):
    """ASCII ply with N vertices, optionally one colour for all of them; same header and line layout as
    ref:bundle_adjust/loader.py:384-406 (what `save_corrected_points` writes as pts3d_adj.ply)."""
    import numpy as np

    pts = np.asarray(point_cloud)
    coloured = not all(c is None for c in color)
    lines = ["ply", "format ascii 1.0", "element vertex {}".format(pts.shape[0]), "property float x", "property float y", "property float z"]
    if coloured:
        lines += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha", "element face 0",
                  "property list uchar int vertex_indices"]
    lines.append("end_header")
    tail = " {} {} {} 255".format(*color) if coloured else ""
    with open(filename, "w") as f:
        f.write("\n".join(lines) + "\n")
        f.writelines("{} {} {}{}\n".format(p[0], p[1], p[2], tail) for p in pts)


def read_point_cloud_ply(filename):
    """The vertices of a ply written by write_point_cloud_ply (ref:bundle_adjust/loader.py:363-381)."""
    import numpy as np

    with open(filename) as f:
        lines = f.read().splitlines()
    start = lines.index("end_header") + 1
    return np.array([[float(v) for v in ln.split()[:3]] for ln in lines[start:] if ln.strip()], dtype=np.float64).reshape(-1, 3)


def save_estimated_params(out_dir, cam_ids, estimated_params):
    """cam_params/<id>.params: for every key of a camera's dict (R, T, C as BundleAdjustmentParameters.reconstruct_vars fills them) the
    key on one line and its values with 16 decimals on the next (ref:bundle_adjust/ba_pipeline.py:606-620)."""
    import os

    for cam_id, params in zip(cam_ids, estimated_params):
        path = "{}/cam_params/{}.params".format(out_dir, cam_id)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            for k in params.keys():
                f.write("{}\n".format(k))
                f.write(" ".join(["{:.16f}".format(v) for v in params[k]]))
                f.write("\n")


def save_dict_to_json(input_dict, output_json_fname):
    """ref:bundle_adjust/loader.py:74-79 (indent 2)."""
    import json

    with open(output_json_fname, "w") as f:
        json.dump(input_dict, f, indent=2)


def load_dict_from_json(input_json_fname):
    """ref:bundle_adjust/loader.py:82-88."""
    import json

    with open(input_json_fname) as f:
        return json.load(f)


def save_projection_matrices(filenames, projection_matrices, crop_offsets):
    """
    P_init/<id>_pinhole.json and P_adj/<id>_pinhole_adj.json of the pipeline (ref:bundle_adjust/loader.py:255-268, written by
    ref:bundle_adjust/ba_pipeline.py:361-378 for affine / perspective runs): the three rows of the 3 x 4 matrix under "P" and the
    crop rectangle of the image as height, width, col_offset, row_offset.
    """
    import os

    import numpy as np

    for fn, P, offset in zip(filenames, projection_matrices, crop_offsets):
        os.makedirs(os.path.dirname(fn) or ".", exist_ok=True)
        P = np.asarray(P, dtype=np.float64)
        save_dict_to_json({"P": [P[0, :].tolist(), P[1, :].tolist(), P[2, :].tolist()],
                           "height": int(offset["height"]), "width": int(offset["width"]),
                           "col_offset": int(offset["col0"]), "row_offset": int(offset["row0"])}, fn)


def load_projection_matrix(filename):
    """(P / P[2, 3], crop offset) of a file written by save_projection_matrices (ref:bundle_adjust/loader.py:271-300)."""
    import numpy as np

    d = load_dict_from_json(filename)
    P = np.array(d["P"], dtype=np.float64)
    return P / P[2, 3], {"col0": d["col_offset"], "row0": d["row_offset"], "width": d["width"], "height": d["height"]}
