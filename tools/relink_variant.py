"""Development aid: relink visfd_amd/_variants/<name>.so from its own recompiled objects plus the
current objects of everything else (after the main library changed)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import build as B  # noqa: E402

name = sys.argv[1]
B.build(verbose=False)
vdir = os.path.join(ROOT, "visfd_amd", "_variants")
units = [(s, os.path.splitext(s)[0], []) for s in B.SOURCES] + B.VARIANTS
objs = []
for s, stem, fl in units:
    v = os.path.join(vdir, "%s_%s.o" % (name, stem))
    objs.append(v if os.path.exists(v) else os.path.join(B.OBJDIR, stem + ".o"))
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(vdir, name + ".so")] + objs)
print("relinked", name)
