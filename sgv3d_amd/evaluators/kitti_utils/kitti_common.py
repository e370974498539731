"""KITTI label files -> annotation dicts, as evaluators/kitti_utils/kitti_common.py:561-669 builds them.

One text line per object: ``name truncated occluded alpha x1 y1 x2 y2 h w l x y z rotation_y [score]``.  Names go through
the reference's ``category_map`` (:10: ``Bus`` counts as ``Car``; any other name is a KeyError there and here);
``dimensions`` are reordered from the file's (h, w, l) to (l, h, w) (:587-590)."""
import pathlib
import re

import numpy as np

__all__ = ['category_map', 'get_label_anno', 'get_label_annos', 'get_image_index_str']

category_map = {'Car': 'Car', 'Bus': 'Car', 'Pedestrian': 'Pedestrian', 'Cyclist': 'Cyclist'}


def get_image_index_str(img_idx):
    return "{:06d}".format(img_idx)


def get_label_anno(label_path):
    with open(label_path, 'r') as f:
        lines = f.readlines()
    rows = [] if (len(lines) == 0 or len(lines[0]) < 15) else [ln.strip().split(' ') for ln in lines]
    num = np.array([[float(v) for v in r[1:15]] for r in rows], np.float64).reshape(-1, 14)
    names = np.array([category_map[r[0]] for r in rows])
    n_real = sum(1 for r in rows if r[0] != 'DontCare')
    anno = {
        'name': names,
        'truncated': num[:, 0].copy(),
        'occluded': num[:, 1].copy(),
        'alpha': num[:, 2].copy(),
        'bbox': num[:, 3:7].copy(),
        'dimensions': num[:, [9, 7, 8]].copy(),                 # file order h, w, l -> l, h, w
        'location': num[:, 10:13].copy(),
        'rotation_y': num[:, 13].copy(),
    }
    if rows and len(rows[0]) == 16:
        anno['score'] = np.array([float(r[15]) for r in rows])
    else:
        anno['score'] = np.zeros((len(rows),))
    anno['index'] = np.array(list(range(n_real)) + [-1] * (len(rows) - n_real), dtype=np.int32)
    anno['group_ids'] = np.arange(len(rows), dtype=np.int32)
    return anno


def get_label_annos(label_folder, image_ids=None, return_ids=False):
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        pat = re.compile(r'^\d{6}.txt$')
        image_ids = sorted(int(p.stem) for p in folder.glob('*.txt') if pat.match(p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    annos = []
    for idx in image_ids:
        anno = get_label_anno(folder / (get_image_index_str(idx) + '.txt'))
        anno["image_idx"] = np.array([idx] * anno["name"].shape[0], dtype=np.int64)
        annos.append(anno)
    return (annos, image_ids) if return_ids else annos
