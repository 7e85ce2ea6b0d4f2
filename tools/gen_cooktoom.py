"""Write monorec_amd/csrc/cooktoom_1d.h (transform code + G tables of the Cook-Toom forms, see monorec_amd/cooktoom.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from monorec_amd import cooktoom  # noqa: E402

if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(cooktoom.__file__)), "csrc", "cooktoom_1d.h")
    with open(path, "w") as f:
        f.write(cooktoom.generate_header() + "\n")
    print(path)
