"""The Ada side (ada/) cannot be compiled in this image: scripts/check_ada_sources.py checks it textually against the
reference's specs instead -- every name used is declared, every `Package.Name` names something that package declares.
Needs the reference tree (absent on the GPU box: skipped there)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "madarch")), reason="no reference tree here")


def _checker():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_ada_sources
    return check_ada_sources


def test_every_name_the_ada_sources_use_is_declared():
    assert _checker().check(REF) == []


def test_the_checker_sees_a_call_of_a_subprogram_that_does_not_exist(tmp_path):
    """... such as the Scenes.Describe / Scenes.Kind_Index of the first round's body, which Madarch.Scenes never had"""
    ada = tmp_path / "ada"
    shutil.copytree(os.path.join(ROOT, "ada"), ada)
    body = (ada / "madarch-renderers.adb").read_text()
    assert "Scenes.HIP.Describe (Scene, Desc, Keep);" in body
    (ada / "madarch-renderers.adb").write_text(body.replace("Scenes.HIP.Describe (Scene, Desc, Keep);", "Scenes.Describe (Scene, Desc);")
                                               .replace("Scenes.HIP.Kind_Index (Self.Scene, Lit)", "Scenes.HIP.Index_Of_Kind (Self.Scene, Lit)"))
    found = _checker().check(REF, str(ada))
    assert any("`Scenes` declares no `describe`" in f for f in found), found
    assert any("index_of_kind" in f.lower() for f in found), found


def test_patches_apply_to_the_reference(tmp_path):
    import subprocess
    subprocess.check_call(["bash", os.path.join(ROOT, "ada", "apply_patches.sh"), REF, str(tmp_path)])
    assert "Max_Dist : GL.Types.Single" in (tmp_path / "madarch-scenes.ads").read_text()
    assert "Max_Dist => Max_Dist," in (tmp_path / "madarch-scenes.adb").read_text()
    spec = (tmp_path / "madarch-renderers.ads").read_text()
    assert "Handle : System.Address" in spec and "Screen_Pass" not in spec
