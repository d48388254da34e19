#!/bin/bash
exec bash $GRAFT_REPO_ROOT/scratch/profile/profile_r04_all.sh
