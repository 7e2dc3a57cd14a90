"""robocupvision_amd: the ROBO-UNet / U-Net training hot path of szemenyeim/RoboCupVision as
hand-written HIP kernels for MI355X (gfx950), behind the reference's nn.Module surface.

    from robocupvision_amd.model import ROBO_UNet, CrossEntropyLoss2d      # drop-in for `from model import *`
"""
__version__ = "0.1.0"
