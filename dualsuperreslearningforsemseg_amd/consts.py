"""Constants of the reference's consts.py:1-2 (restated as data)."""
NUM_RGB_CHANNELS = 3
IMAGE_FILE_EXTENSIONS = ('.png', '.jpg', '.jpeg', '.bmp')
