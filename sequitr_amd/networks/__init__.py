"""Networks of the hot path: the U-Net (unet.py) and the progressive WGAN-GP (gan.py), plus the TF-style variable
scopes both use (scope.py)."""
