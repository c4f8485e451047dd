"""TEST INFRASTRUCTURE ONLY -- empty stand-in so `import cv2` in the reference succeeds."""
