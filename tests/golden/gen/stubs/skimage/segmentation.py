"""Restatement (ours) of scikit-image 0.19.3 find_boundaries (third-party, absent offline).
Parity at this call site (pursuit_env.py:20) is "unpinned": no reference fixture covers it."""
import numpy as np
from scipy import ndimage as ndi


def find_boundaries(label_img, connectivity=1, mode='thick', background=0):
    if label_img.dtype == 'bool':
        label_img = label_img.astype(np.uint8)
    fp = ndi.generate_binary_structure(label_img.ndim, connectivity)
    b = ndi.grey_dilation(label_img, footprint=fp) != ndi.grey_erosion(label_img, footprint=fp)
    if mode == 'inner':
        b &= (label_img != background)
    return b
