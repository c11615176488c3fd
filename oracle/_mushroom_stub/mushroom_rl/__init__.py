"""Duck-typed stand-in for the parts of MushroomRL the reference's ATACOM core imports.

Used ONLY by oracle/gen_golden.py inside the build container so that the reference's own modules
(/root/reference/atacom/...) can be imported unchanged to capture golden vectors.  MushroomRL is a
third-party dependency of the reference (requirements.txt: mushroom-rl>=1.7.0) that is not vendored
and not installed here.  This is our code, not MushroomRL's; it implements just the attribute
surface the reference touches.
"""
