"""Empty stand-in: no plotting on the build host."""
