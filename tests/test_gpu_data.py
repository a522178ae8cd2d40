"""GPU: input-pipeline colour conversions (csrc/colour.hip behind srcgan_amd.data) against the oracle's float64 restatement
of dataset.py:92-159 (scikit-image is absent: parity unpinned, see oracle/srcgan_oracle.py) and through the 8-bit round trip."""
import numpy as np
import pytest
import torch

from oracle import srcgan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _images(B, H, W, seed):
    rng = np.random.default_rng(seed)
    im = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    im[0, 0, :6] = [[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 10, 10]]   # both branches of every piecewise map
    return im


@pytest.mark.parametrize("shape", [(1, 8, 8), (3, 37, 53), (2, 128, 96)])
def test_conversions_match_oracle(shape):
    from srcgan_amd import data as D
    B, H, W = shape
    im = _images(B, H, W, 5)
    dev = torch.from_numpy(im).cuda()
    for name in ("arr2gray", "arr2rgb", "arr2lab", "arr2ab"):
        got = getattr(D, name)(dev).cpu()
        ref = torch.stack([getattr(O, name)(im[b]) for b in range(B)])
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert float((got - ref).abs().max()) <= 2e-7, name          # double arithmetic on both sides, one rounding to float
    one = D.arr2lab(dev[0])                                           # single image form ([H,W,3] -> [3,H,W])
    assert torch.equal(one, D.arr2lab(dev)[0])
    sample = D.G2LAB()(dev, dev)
    assert torch.equal(sample["src"], D.arr2gray(dev)) and torch.equal(sample["tar"], D.arr2lab(dev))
    sample = D.G2RGB()(dev, dev)
    assert torch.equal(sample["tar"], D.arr2rgb(dev))


def test_lab_round_trip_and_inverse():
    """rgb -> normalised LAB -> 8-bit rgb: within one code value of the input (truncating store, as dataset.py:101), equal to the
    oracle's inverse except where the float64 product lands within 1e-6 of an integer; ab2img == lab2img on the split planes."""
    from srcgan_amd import data as D
    im = _images(2, 64, 80, 9)
    dev = torch.from_numpy(im).cuda()
    lab = D.arr2lab(dev)
    back = D.lab2img(lab)
    assert back.dtype == torch.uint8 and back.shape == dev.shape
    assert int((back.int() - dev.int()).abs().max()) <= 1
    ref = np.stack([O.lab2img(lab[b].cpu().numpy().transpose(1, 2, 0)) for b in range(2)])
    assert (back.cpu().numpy() != ref).mean() < 1e-3
    assert torch.equal(D.ab2img(lab[:, :1], lab[:, 1:]), back)


def test_bad_inputs_are_refused():
    from srcgan_amd import data as D
    with pytest.raises(ValueError):
        D.arr2gray(torch.zeros(4, 4, 3, device="cuda"))
    with pytest.raises(ValueError):
        D.arr2rgb(torch.zeros(4, 4, 4, dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        D.lab2img(torch.zeros(2, 4, 4, device="cuda"))


def test_folder_dataset_device_batches(tmp_path):
    """dataset.G2LAB / G2RGB behind a DataLoader (testCas.py:44,61-62): files on disk -> the {'src','tar','idx'} batch on the
    device, equal to the oracle's per-sample ``__getitem__`` maths on the decoded arrays; ``show`` writes the framed pair."""
    from PIL import Image
    from srcgan_amd import data as D
    from test_host_logic import _make_folder_dataset
    names, arrays = _make_folder_dataset(tmp_path, n=5, hw=(12, 16), up=2)
    for ver, conv in (("G2LAB", O.arr2lab), ("G2RGB", O.arr2rgb)):
        train, _, test = D.load_dataset("Mini", ver, dataset_dir=str(tmp_path))
        seen = 0
        for batch in D.DeviceLoader(train, 2, num_workers=2, shuffle=False, drop_last=False):
            assert batch["src"].is_cuda and batch["src"].shape[1:] == (1, 12, 16) and batch["tar"].shape[1:] == (3, 24, 32)
            for j, idx in enumerate(batch["idx"].tolist()):
                src, tar = arrays[names[idx]]
                assert float((batch["src"][j].cpu() - O.arr2gray(src)).abs().max()) <= 2e-7
                assert float((batch["tar"][j].cpu() - conv(tar)).abs().max()) <= 2e-7
                seen += 1
        assert seen == 3
        path = test.show(0, save_dir=str(tmp_path / "example"))
        img = np.asarray(Image.open(path))
        assert img.shape == (24 + 10, 2 * (32 + 10), 3) and (img[:5] == 255).all() and (img[:, :5] == 255).all()
        if ver == "G2RGB":                                                       # right half = the target inside its frame
            assert np.array_equal(img[5:-5, 42 + 5:-5], arrays[names[4]][1])
