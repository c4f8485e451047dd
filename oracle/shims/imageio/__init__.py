"""TEST INFRASTRUCTURE ONLY -- empty stand-in so `import imageio` in the reference succeeds."""
from . import v3  # noqa: F401
