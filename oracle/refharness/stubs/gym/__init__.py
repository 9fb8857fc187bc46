# Import-only stand-in so the read-only Python reference can be imported in the
# build container (gym==0.14.0 is not installed).  Test infrastructure only;
# never shipped, never imported by the product path.
