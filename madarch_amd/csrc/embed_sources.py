"""Writes mdh_jit_sources.inc: the device headers as C++ raw string literals (split below the
64 KiB some compilers allow per literal; adjacent literals concatenate)."""
import os
here = os.path.dirname(os.path.abspath(__file__))
for macro, name in (("MDH_SRC_DEVICE", "mdh_device.h"), ("MDH_SRC_MARCH", "mdh_march.h"), ("MDH_SRC_KERNELS", "mdh_kernels.h")):
    text = open(os.path.join(here, name)).read()
    assert ")MDHSRC\"" not in text
    print("static const char %s[] =" % macro)
    lines, chunk = text.splitlines(keepends=True), ""
    for line in lines:
        if len(chunk) + len(line) > 16000:
            print('R"MDHSRC(' + chunk + ')MDHSRC"')
            chunk = ""
        chunk += line
    print('R"MDHSRC(' + chunk + ')MDHSRC";')
