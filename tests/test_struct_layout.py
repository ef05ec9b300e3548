"""ctypes mirrors of the kernel argument structs against the compiled library (runs without a GPU: the
library loads, only launches need a device)."""
import ctypes as C

import pytest

from d4pg_pytorch_amd.ops import hip as H


def test_every_mirrored_struct_is_checked_against_the_library():
    lib = H.lib()
    names = {c.__name__ for c in H._mirrored_structs()}
    # every launch descriptor of every kernel family is in the set
    assert {"GemmGroup", "D4pgLossArgs", "D3pgLossArgs", "AdamPolyakArgs", "ShadowArgs", "InsertArgs", "SampleArgs",
            "SampleRowsArgs", "TreeUpdateArgs", "UnpackArgs", "NstepArgs", "OuArgs", "CriticPhaseArgs", "ActorPhaseArgs",
            "WgradGroup", "CriticPhaseArgs32", "ActorPhaseArgs32", "WgradGroup32", "ActorInferArgs32",
            "PackActorArgs32", "PendulumArgs", "RolloutRowsArgs", "PhaseDims", "LayerW", "LayerW32"} <= names
    assert H.check_struct_sizes(lib) == len(names) >= 40


def test_a_drifted_mirror_is_refused(monkeypatch):
    lib = H.lib()

    class NstepArgs(C.Structure):         # two fields short of csrc/nstep.hip
        _fields_ = H.NstepArgs._fields_[:-2]
    NstepArgs.__module__ = H.__name__
    monkeypatch.setattr(H, "NstepArgs", NstepArgs)
    with pytest.raises(RuntimeError, match="NstepArgs"):
        H.check_struct_sizes(lib)


def test_a_mirror_without_an_export_is_refused(monkeypatch):
    class BrandNewArgs(C.Structure):
        _fields_ = [("x", C.c_int)]
    BrandNewArgs.__module__ = H.__name__
    monkeypatch.setattr(H, "BrandNewArgs", BrandNewArgs, raising=False)
    with pytest.raises(AttributeError):
        H.check_struct_sizes(H.lib())
