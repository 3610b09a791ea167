"""Reference module location -> gen3c_amd.gen3c_persistent (the resident model behind the GUI / API server, gen3c_persistent.py:55-569)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from gen3c_amd.gen3c_persistent import *  # noqa: E402,F401,F403
from gen3c_amd.gen3c_persistent import Gen3cPersistentModel  # noqa: E402,F401
