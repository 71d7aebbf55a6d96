"""Import-path twin of the reference's SWIG output directory `cpp_core/pcd_tiling/build/` (generate_wraper.sh)."""
