from mfp.main import main

main()
