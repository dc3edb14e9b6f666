"""TEST SCAFFOLDING -- a stand-in for the `lightning` package, which is not installed in the build image (no network).
Only tests/test_lightning_surface.py puts this directory on sys.path, in a child process."""
