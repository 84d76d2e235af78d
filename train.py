#!/usr/bin/env python3
"""`python train.py <flags>` from the repository root, as bash_scripts/run_joint.sh:285-326 invokes the
reference."""
from cooperativeimagecaptioning_amd.train import main

if __name__ == '__main__':
    main()
