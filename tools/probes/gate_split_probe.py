#!/usr/bin/env python3
"""Would the decode's gate product be cheaper as TWO K = 512 products on the LDS-staged logit walker (64-row groups: weights read
4x instead of 8x) than as one K = 1024 product on the 32-row strip walker?  Times the three launches back to back."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import step_gemms as sg  # noqa: E402

for sh in [('i2h+h2h (one K=1024 launch)', 256, 2560, 512, 512, 1, 0), ('x i2h', 256, 2560, 512, 0, 1, 0),
           ('h [h2h; h2att]', 256, 3072, 512, 0, 1, 0), ('h h2h', 256, 2560, 512, 0, 1, 0)]:
    us, err = sg.run(*sh, flag=1)
    print(f'{sh[0]:32s} M{sh[1]} N{sh[2]} K{sh[3] + sh[4]}: {us:7.2f} us  err {err:.1e}')
