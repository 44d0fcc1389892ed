"""Empty stand-in: the golden-vector generator never samples."""
