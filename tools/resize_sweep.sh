for spec in "-" "-:SARPRO_HIP_RESIZE_BLOCKS=1024" "-:SARPRO_HIP_RESIZE_BLOCKS=4096" "-:SARPRO_HIP_RESIZE_BLOCKS=8192" "lib_rh1.so" "lib_rh1.so:SARPRO_HIP_RESIZE_BLOCKS=4096" "lib_rh1.so:SARPRO_HIP_RESIZE_BLOCKS=8192"; do
  lib=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  ( [ "$lib" != "-" ] && export SARPRO_HIP_LIB=$PWD/sarpro_amd/$lib; [ -n "$envs" ] && export $envs; echo -n "$spec  "; python tools/time_resize_flow.py 2>&1 | grep register | sed 's/.*resize_h/resize_h/' )
done
