#!/bin/bash
HP_TAG="default" python tools/host_path_modes.py 2>/dev/null | tail -1
HP_TAG="raw arrays NOT freed" HP_KEEP_RAW=1 python tools/host_path_modes.py 2>/dev/null | tail -1
HP_TAG="raw arrays NOT freed #2" HP_KEEP_RAW=1 python tools/host_path_modes.py 2>/dev/null | tail -1
