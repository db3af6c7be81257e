"""CPU: weightmap.py surface: ImageLabels and the create_weightmaps
folder walker / file naming, numerically equal to ImageWeightMap2 on the binarised labels."""
import os

import numpy as np
import pytest

from sequitr_amd import weightmap as wm
from sequitr_amd.pipeline import ImageWeightMap2

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.npz"))


def test_image_labels_2d_and_stack():
    lab = (G["wm_in_0"] * 7).astype(np.uint16)
    l = wm.ImageLabels(lab)
    assert l.outputs == 2 and l.labels().dtype == np.uint8 and set(np.unique(l.labels())) <= {0, 1}
    stack = np.zeros((2, 8, 8), np.uint8)
    stack[0, :4] = 5
    stack[1, 6:] = 1
    l = wm.ImageLabels(stack)
    assert l.outputs == 3 and l.labels()[0, 0] == 1 and l.labels()[7, 0] == 2 and l.labels()[5, 0] == 0
    with pytest.raises(ValueError):
        wm.ImageLabels(np.ones((5, 4, 4), np.uint8))
    with pytest.raises(AssertionError):
        wm.ImageLabels(np.ones(4))


def test_names():
    assert wm.weights_folder_name(30., 3.) == "weights_w0-30.00_sigma-3.00"
    assert wm.weights_folder_name(30., 3., False) == "weights"
    assert wm.weights_file_name("0001_pos3_label.tif") == "0001_pos3_weights.tif"
    assert wm.weights_file_name("cellA_label.tif") == "cellA_weights.tif"


def test_create_weightmaps_walks_folders_and_matches_reference_arrays(tmp_path):
    for d in ("set1", "set2"):
        os.makedirs(str(tmp_path / d / "label"))
    wm.imsave(str(tmp_path / "set1" / "label" / "0001_a_label.tif"), (G["wm_in_0"] * 255).astype(np.uint8))
    wm.imsave(str(tmp_path / "set1" / "label" / "0002_a_label.tif"), (G["wm_in_1"] * 255).astype(np.uint8))
    wm.imsave(str(tmp_path / "set2" / "label" / "0001_b_label.tif"), (G["wm_in_2"] * 255).astype(np.uint8))
    open(str(tmp_path / "set2" / "label" / "notes.txt"), "w").write("ignored")
    out = wm.create_weightmaps(str(tmp_path), ["set1", "set2"], w0=10., sigma=5.)
    assert [os.path.relpath(o, str(tmp_path)) for o in out] == [
        "set1/weights_w0-10.00_sigma-5.00/0001_a_weights.tif", "set1/weights_w0-10.00_sigma-5.00/0002_a_weights.tif",
        "set2/weights_w0-10.00_sigma-5.00/0001_b_weights.tif"]
    for o, key in zip(out, ("wm2_out_0", "wm2_out_1", "wm2_out_2")):
        got = wm.imread(o)
        assert got.dtype == np.float32 and got.shape == (64, 64)
        # the reference's ImageWeightMap2 output for the same labels (golden), stored as float32
        assert np.allclose(got, np.squeeze(G[key]).astype(np.float32), rtol=1e-6, atol=1e-6)
    w = ImageWeightMap2(10., 5.)(G["wm_in_0"].astype(bool))
    assert np.allclose(np.squeeze(w), np.squeeze(G["wm2_out_0"]), rtol=1e-12)
