"""Reference entry-point location -> gen3c_amd.gen3c_single_image (same flags; works as a script path under python / torchrun and as
`-m cosmos_predict1.diffusion.inference.gen3c_single_image`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from gen3c_amd.gen3c_single_image import create_parser, demo, main  # noqa: E402,F401

if __name__ == "__main__":
    main()
