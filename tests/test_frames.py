"""Video-frame transform: csrc/frames.hip through the reference's processor surface, on the CPU lane emulator here and on MI355X
under -m gpu, against oracle/frames.py."""
import os
import subprocess

import pytest
import torch

import frame_cases as fc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIB = os.path.join(ROOT, "tests", "emu", "build", "libantmmf_emu.so")


@pytest.fixture(scope="module")
def emu():
    from test_kernels_emu import _stale

    if _stale():
        subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    from antmmf.hip import _lib

    old = os.environ.get("ANTMMF_HIP_LIB")
    os.environ["ANTMMF_HIP_LIB"] = EMU_LIB
    _lib.reset_for_tests()
    yield torch.device("cpu")
    if old is None:
        os.environ.pop("ANTMMF_HIP_LIB", None)
    else:
        os.environ["ANTMMF_HIP_LIB"] = old
    _lib.reset_for_tests()


def test_oracle_resize_size_and_scales():
    from oracle import frames as of

    assert of.resize_size(360, 640, 448) == (252, 448) and of.resize_size(640, 360, 448) == (448, 252) and of.resize_size(50, 50, 224) == (224, 224)
    assert of.scales(448, True) == [224, 256, 288, 320, 352, 384, 416, 448] and of.scales(300, True) == [224, 256, 288, 300] and of.scales(448, False) == [448]


def test_processor_emulated(emu):
    print(fc.case_processor(emu))


def test_antialias_emulated(emu):
    print(fc.case_antialias(emu, shapes=((2, 48, 64, 24), (2, 64, 48, 40), (1, 33, 57, 64), (1, 97, 131, 32), (2, 20, 30, 48), (1, 7, 5, 3))))


def test_dark_clip_emulated(emu):
    fc.case_dark_clip_keeps_reference_semantics(emu)


def test_collate_emulated(emu):
    assert fc.case_collate(emu, n_clips=1, num_frm=1)[0] == 3   # (one frame per video: a 224-px frame is ~150 k emulated threads)


@pytest.mark.gpu
def test_processor_gpu():
    print(fc.case_processor(torch.device("cuda:0")))


@pytest.mark.gpu
def test_antialias_gpu():
    print(fc.case_antialias(torch.device("cuda:0")))


@pytest.mark.gpu
def test_dark_clip_and_collate_gpu():
    dev = torch.device("cuda:0")
    fc.case_dark_clip_keeps_reference_semantics(dev)
    fc.case_collate(dev)


@pytest.mark.gpu
def test_full_size_video_gpu():
    assert fc.case_full_size(torch.device("cuda:0")) == (252, 448)
