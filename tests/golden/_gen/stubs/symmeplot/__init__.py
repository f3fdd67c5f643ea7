"""Import stand-in for ``symmeplot`` (3-D animation helper some gallery
scripts import at the top; never reached by ``gallery_capture.py``, which
stops a script at ``Problem.solve``).  Build-container only."""
