"""Empty stand-in: the reference only needs `import tensorflow.compat.v1 as tf` to succeed."""
