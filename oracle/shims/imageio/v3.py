"""TEST INFRASTRUCTURE ONLY -- empty stand-in for imageio.v3."""
